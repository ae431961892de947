function [u0, L0, SV_history] = mf_nn_curvature_registration_no_ref_ml(img, optPara)
% Nuclear Norm Experiments/mf_nn_curvature_registration_no_ref_ml.m:8 over the engine; like the reference it returns
% only the last iterate (:204-206).
[u, L, SV_history] = cp_mf_nn_driver(img, optPara, 'curvature', false);
lev = size(u, 1);
o = optPara.outerIter(min(lev, numel(optPara.outerIter)));
u0 = u{lev, o};
L0 = L{lev, o}(:);
end
