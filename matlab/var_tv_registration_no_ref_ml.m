function u = var_tv_registration_no_ref_ml(img, optPara)
% Nuclear Norm Experiments/var_tv_registration_no_ref_ml.m:7 over the engine.
u = cp_var_driver(img, optPara, 'tv', false);
end
