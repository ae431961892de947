function u = cp_var_driver(img, optPara, reg, single_level)
% Host loop of Nuclear Norm Experiments/var_tv_registration_no_ref_ml.m:40-197 (reg = 'tv'),
% var_curvature_registration_no_ref_ml.m (reg = 'curvature') and var_tv_registration_no_ref.m (single_level = true).
% Tested transcription: ma_amd/drivers.py::_var_ml.
c = cp_const();
k = numel(img);
for i = 1 : k
    img{i} = (img{i} - min(img{i}(:))) / (max(img{i}(:)) - min(img{i}(:)));
end
[m, n] = size(img{1});
omega = [0, m, 0, n];
if single_level
    numLevels = 1;
else
    numLevels = min(floor(log2(m) - 5), floor(log2(n) - 5)) + 1;                    % (:58)
end
if isscalar(optPara.outerIter), optPara.outerIter = optPara.outerIter * [1, 1]; end
u = cell(numLevels, max(optPara.outerIter));
if strcmp(reg, 'tv'), model = c.MODEL_VAR_TV; else, model = c.MODEL_VAR_CURV; end
flags = 0;
if single_level, flags = c.FLAG_VAR_SSD_SWAP; end      % SSD(y1, b, sigma, 1, flag) of var_tv_registration_no_ref.m:173
ms = zeros(numLevels, 1); ns = zeros(numLevels, 1);
ms(numLevels) = m; ns(numLevels) = n;
for lev = numLevels - 1 : -1 : 1
    ms(lev) = ceil(ms(lev + 1) / 2); ns(lev) = ceil(ns(lev + 1) / 2);
end
hdl = cell(numLevels, 1);
cleanup = onCleanup(@() destroy_all(hdl));
for lev = 1 : numLevels
    h_grid = [(omega(2) - omega(1)) / ms(lev), (omega(4) - omega(3)) / ns(lev)];     % (:85)
    % omega = []: these drivers call evaluate_displacement without omega (:147-148) -> its default region
    hdl{lev} = cp_mex('create', model, ms(lev), ns(lev), k, cp_bc_id(optPara.bc), h_grid, [], 0, optPara.mu, 1, flags);
end
stack = zeros(m * n, k);
for i = 1 : k, stack(:, i) = img{i}(:); end
cp_mex('set_images', hdl{numLevels}, stack(:));
for lev = numLevels : -1 : 2
    cp_mex('downsample', hdl{lev}, hdl{lev - 1});      % (:71-74)
end
for lev = 1 : numLevels
    outerIter = optPara.outerIter(min(lev, 2));
    if lev > 1
        cp_mex('prolong', hdl{lev - 1}, hdl{lev}, 2);  % 2 * kron(x, ones(2)) (:109-114)
        cp_mex('destroy', hdl{lev - 1}); hdl{lev - 1} = [];
    end
    mn = ms(lev) * ns(lev);
    for o = 1 : outerIter
        cp_mex('linearize', hdl{lev});                 % (:141-161)
        norm_A_est = cp_mex('norm_estimate', hdl{lev});                              % (:164-165)
        tau = sqrt(0.99 / norm_A_est ^ 2);             % (:168-169)
        sigma = sqrt(0.99 / norm_A_est ^ 2);
        [~, reason] = cp_mex('solve', hdl{lev}, optPara.theta, tau, sigma, optPara.maxIter, optPara.tol);
        [ph, dh] = cp_mex('get_history', hdl{lev});
        cp_print_log('F', 'G', optPara.theta, tau, sigma, 2 * k * mn, numel(cp_mex('get_dual', hdl{lev})), ...
            optPara.maxIter, optPara.tol, ph, dh, reason == c.STOP_TOLERANCE);
        u{lev, o} = reshape(cp_mex('get_primal', hdl{lev}), mn, 2, k);               % (:179-182)
    end
end
end

function destroy_all(hdl)
for i = 1 : numel(hdl)
    if ~isempty(hdl{i})
        try, cp_mex('destroy', hdl{i}); catch, end
    end
end
end
