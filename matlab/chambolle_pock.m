function [x_star, y_star, primal_history, dual_history] = ...
    chambolle_pock(F, G, K, x0, y0, theta, tau, sigma, maxIter, tol)
% Same signature, defaults and outputs as Primal Dual Optimization/chambolle_pock.m:1-2,54-70.
%   F, G operator structs (TV_registration([], sigma), SAD_registration([], u0, T, R, h, lambda, tau), ...):
%       the whole iteration runs in libcpengine.so through cp_mex; histories come back with the reference's columns.
%   F, G function handles: arbitrary closures cannot be fused; the original loop runs (the reference's own chambolle_pock.m, found on the path by cp_call_shadowed).
if ~isstruct(F) || ~isstruct(G)
    args = {F, G, K, x0, y0};
    if nargin >= 6,  args{end + 1} = theta;   end
    if nargin >= 7,  args{end + 1} = tau;     end
    if nargin >= 8,  args{end + 1} = sigma;   end
    if nargin >= 9,  args{end + 1} = maxIter; end
    if nargin >= 10, args{end + 1} = tol;     end
    [x_star, y_star, primal_history, dual_history] = cp_call_shadowed('chambolle_pock', args{:});
    return
end
c = cp_const();

% problem size and model from the (F, G, K) triple
switch [F.type, '+', G.type]
    case 'TV_registration+SAD_registration', model = c.MODEL_TV_L1_REG;     [m, n] = size(G.T); h = G.h; ncomp = 2;
    case 'TV_registration+SSD_registration', model = c.MODEL_TV_L2_REG;     [m, n] = size(G.T); h = G.h; ncomp = 2;
    case 'TV_denoise+SAD_denoise'
        if ~isstruct(K), error('cpengine:badArg', 'denoising needs K = cp_gradient_operator(m, n, h, 1)'); end
        model = c.MODEL_TV_L1_DENOISE; m = K.m; n = K.n; h = K.h; ncomp = 1;
    otherwise
        error('cpengine:unsupported', 'unsupported (F, G) combination %s + %s', F.type, G.type);
end
if isstruct(K)
    if K.m ~= m || K.n ~= n || K.ncomp ~= ncomp, error('cpengine:badArg', 'K does not match G'); end
    h = K.h;
elseif ~isequal(size(K), [2 * ncomp * m * n, ncomp * m * n])
    error('cpengine:badArg', 'K has the wrong size for these operators');
end

% nargin defaults (:54-70); ||K||_1 * ||K||_inf of the stencil = (2/h1 + 2/h2) * max(2/h1, 2/h2)
if nargin < 10, tol = 1e-3; end
if nargin < 9, maxIter = 300; end
L_squared_estimate = (2 / h(1) + 2 / h(2)) * max(2 / h(1), 2 / h(2));
if nargin < 8 && nargin >= 7, sigma = 1 / (L_squared_estimate * tau); end
if nargin < 7
    tau = 1 / sqrt(L_squared_estimate);
    sigma = tau;
end
if nargin < 6, theta = 1; end

hdl = cp_mex('create', model, m, n, 1, c.BC_NEUMANN, h, [], G.lambda, 0);
cleanup = onCleanup(@() cp_mex('destroy', hdl));
if model == c.MODEL_TV_L1_DENOISE
    cp_mex('set_images', hdl, G.g);
else
    cp_mex('set_images', hdl, G.T(:), G.R(:));
    cp_mex('linearize', hdl, G.u0);                 % evaluate_displacement at u0 (SAD_registration.m:39-40), once
end
cp_mex('set_primal', hdl, x0(:));
cp_mex('set_dual', hdl, y0(:));
[~, reason] = cp_mex('solve', hdl, theta, tau, sigma, maxIter, tol);
x_star = cp_mex('get_primal', hdl);
y_star = cp_mex('get_dual', hdl);
[primal_history, dual_history] = cp_mex('get_history', hdl);
cp_print_log(F.type, G.type, theta, tau, sigma, numel(x0), numel(y0), maxIter, tol, ...
    primal_history, dual_history, reason == c.STOP_TOLERANCE);
end
