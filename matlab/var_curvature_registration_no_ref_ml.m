function u = var_curvature_registration_no_ref_ml(img, optPara)
% Nuclear Norm Experiments/var_curvature_registration_no_ref_ml.m:7 over the engine. The Laplacian's boundary condition
% comes from optPara.bc (the reference's call at :119 omits the argument).
u = cp_var_driver(img, optPara, 'curvature', false);
end
