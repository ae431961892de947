function [u, L, SV_history] = mf_nn_tv_registration_no_ref(img, optPara)
% Nuclear Norm Experiments/mf_nn_tv_registration_no_ref.m over the engine: one level, F2 = mu * norm21 (:209-216).
[u, L, SV] = cp_mf_nn_driver(img, optPara, 'tv', true);
u = u(1, :)'; L = L(1, :)'; SV_history = SV{1};
end
