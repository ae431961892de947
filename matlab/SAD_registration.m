function [res1, res2, res3] = SAD_registration(u, u0, T, R, h, lambda, tau, conjugate_flag)
% Same signature as Primal Dual Optimization/SAD_registration.m:1-2.
%   G = SAD_registration([], u0, T, R, h, lambda, tau)   operator struct for chambolle_pock (engine path); replaces
%                                     G = @(u, c_flag) SAD_registration(u, u0, T, R, h, lambda, tau, c_flag)
%   numeric u: the reference's own SAD_registration.m (cp_call_shadowed)
if isempty(u)
    res1 = struct('type', 'SAD_registration', 'u0', u0(:), 'T', T, 'R', R, 'h', h(:)', 'lambda', lambda, 'tau', tau);
    return
end
if nargin < 8, conjugate_flag = false; end
if nargout == 3
    [res1, res2, res3] = cp_call_shadowed('SAD_registration', u, u0, T, R, h, lambda, tau, conjugate_flag);
else
    [res1, res2] = cp_call_shadowed('SAD_registration', u, u0, T, R, h, lambda, tau, conjugate_flag);
end
end
