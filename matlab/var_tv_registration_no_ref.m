function u = var_tv_registration_no_ref(img, optPara)
% Nuclear Norm Experiments/var_tv_registration_no_ref.m over the engine, incl. its swapped SSD prox arguments (:173).
u = cp_var_driver(img, optPara, 'tv', true);
u = u(1, :)';
end
