function [u, L, SV_history] = mf_nn_tv_registration_no_ref_ml(img, optPara)
% Nuclear Norm Experiments/mf_nn_tv_registration_no_ref_ml.m:8 over the engine.
[u, L, SV_history] = cp_mf_nn_driver(img, optPara, 'tv', false);
end
