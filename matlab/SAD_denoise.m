function [res1, res2, res3] = SAD_denoise(u, g, lambda, tau, conjugate_flag)
% Same signature as Primal Dual Optimization/SAD_denoise.m:1.
%   G = SAD_denoise([], g, lambda, tau)   operator struct for chambolle_pock (engine path)
%   numeric u: the reference's own SAD_denoise.m (cp_call_shadowed)
if isempty(u)
    res1 = struct('type', 'SAD_denoise', 'g', g(:), 'lambda', lambda, 'tau', tau);
    return
end
if nargin < 5, conjugate_flag = false; end
if nargout == 3
    [res1, res2, res3] = cp_call_shadowed('SAD_denoise', u, g, lambda, tau, conjugate_flag);
else
    [res1, res2] = cp_call_shadowed('SAD_denoise', u, g, lambda, tau, conjugate_flag);
end
end
