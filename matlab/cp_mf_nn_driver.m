function [u, L, SV_history] = cp_mf_nn_driver(img, optPara, reg, single_level)
% Host loop of Nuclear Norm Experiments/mf_nn_tv_registration_no_ref_ml.m:40-226 (reg = 'tv'),
% mf_nn_curvature_registration_no_ref_ml.m (reg = 'curvature') and of the single-level predecessor
% mf_nn_tv_registration_no_ref.m (single_level = true); everything inside an outer iteration runs on the device.
% The tested transcription of this sequence is ma_amd/drivers.py::_mf_nn_ml.
c = cp_const();
k = numel(img);
for i = 1 : k
    img{i} = (img{i} - min(img{i}(:))) / (max(img{i}(:)) - min(img{i}(:)));          % normalize (:43)
end
[m, n] = size(img{1});
omega = [0, m, 0, n];
if single_level
    numLevels = 1;
else
    numLevels = min(floor(log2(m) - 5), floor(log2(n) - 5)) + 1;                    % (:59)
end
if isscalar(optPara.outerIter), optPara.outerIter = optPara.outerIter * [1, 1]; end
if isscalar(optPara.nu_factor), optPara.nu_factor = optPara.nu_factor * [1, 1]; end
u = cell(numLevels, max(optPara.outerIter));
L = cell(numLevels, max(optPara.outerIter));
SV_history = cell(numLevels, 1);
if strcmp(reg, 'tv'), model = c.MODEL_MF_NN_TV; else, model = c.MODEL_MF_NN_CURV; end
flags = 0;
if single_level, flags = c.FLAG_MFNN_NORM21; end       % F2 = mu * norm21, SAD weight 1 (mf_nn_tv_registration_no_ref.m:206-216)

% level sizes, coarsest first (:67-79): every level halves with ceil
ms = zeros(numLevels, 1); ns = zeros(numLevels, 1);
ms(numLevels) = m; ns(numLevels) = n;
for lev = numLevels - 1 : -1 : 1
    ms(lev) = ceil(ms(lev + 1) / 2); ns(lev) = ceil(ns(lev + 1) / 2);
end
hdl = cell(numLevels, 1);
cleanup = onCleanup(@() destroy_all(hdl));
for lev = 1 : numLevels
    h_grid = [(omega(2) - omega(1)) / ms(lev), (omega(4) - omega(3)) / ns(lev)];     % (:88)
    hdl{lev} = cp_mex('create', model, ms(lev), ns(lev), k, cp_bc_id(optPara.bc), h_grid, omega, 0, optPara.mu, 1, flags);
end
stack = zeros(m * n, k);
for i = 1 : k, stack(:, i) = img{i}(:); end
cp_mex('set_images', hdl{numLevels}, stack(:));
for lev = numLevels : -1 : 2
    cp_mex('downsample', hdl{lev}, hdl{lev - 1});      % conv2 + subsampling (:75-76)
end

nu = 0;
for lev = 1 : numLevels
    if lev == 1
        outerIter = optPara.outerIter(1); nu_factor = optPara.nu_factor(1);
    else
        outerIter = optPara.outerIter(2); nu_factor = optPara.nu_factor(2);
        cp_mex('prolong', hdl{lev - 1}, hdl{lev}, 1);  % kron(x, ones(2)), dual start 0 (:111-122)
        cp_mex('destroy', hdl{lev - 1}); hdl{lev - 1} = [];
    end
    SV_history{lev} = zeros(k, outerIter);             % (:153)
    mn = ms(lev) * ns(lev);
    for o = 1 : outerIter
        cp_mex('linearize', hdl{lev});                 % evaluate_displacement of all images at u0 (:164-169)
        if o == 1 && lev == 1
            nu = nu_factor * sum(cp_mex('singular_values', hdl{lev}, 0));            % (:173-175)
        elseif o == 1
            nu = 2 * nu_factor * nu;                   % (:177)
        else
            nu = nu_factor * nu;                       % (:179)
        end
        cp_mex('set_nu', hdl{lev}, nu);
        norm_A_est = cp_mex('norm_estimate', hdl{lev});                              % matrix_norm (:191-192)
        tau = sqrt(0.99 / norm_A_est ^ 2);
        sigma = sqrt(0.99 / norm_A_est ^ 2);
        [~, reason] = cp_mex('solve', hdl{lev}, optPara.theta, tau, sigma, optPara.maxIter, optPara.tol);
        [ph, dh] = cp_mex('get_history', hdl{lev});
        cp_print_log('F', 'G', optPara.theta, tau, sigma, 3 * k * mn, numel(cp_mex('get_dual', hdl{lev})), ...
            optPara.maxIter, optPara.tol, ph, dh, reason == c.STOP_TOLERANCE);
        x = cp_mex('get_primal', hdl{lev});
        u{lev, o} = reshape(x(1 : 2 * k * mn), mn, 2, k);                            % (:207-212)
        L{lev, o} = reshape(x(2 * k * mn + 1 : end), ms(lev), ns(lev), k);
        SV_history{lev}(:, o) = cp_mex('singular_values', hdl{lev}, 1);              % (:214-215)
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
