function [res1, res2, res3] = SSD_registration(u, u0, T, R, h, lambda, tau, conjugate_flag)
% Same signature as Primal Dual Optimization/SSD_registration.m:1-2.
%   G = SSD_registration([], u0, T, R, h, lambda, tau)   operator struct for chambolle_pock (engine path); replaces
%                                     G = @(u, c_flag) SSD_registration(u, u0, T, R, h, lambda, tau, c_flag)
%   numeric u: the reference's own SSD_registration.m (cp_call_shadowed)
if isempty(u)
    res1 = struct('type', 'SSD_registration', 'u0', u0(:), 'T', T, 'R', R, 'h', h(:)', 'lambda', lambda, 'tau', tau);
    return
end
if nargin < 8, conjugate_flag = false; end
if nargout == 3
    [res1, res2, res3] = cp_call_shadowed('SSD_registration', u, u0, T, R, h, lambda, tau, conjugate_flag);
else
    [res1, res2] = cp_call_shadowed('SSD_registration', u, u0, T, R, h, lambda, tau, conjugate_flag);
end
end
