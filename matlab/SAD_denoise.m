function [res1, res2, res3] = SAD_denoise(u, g, lambda, tau, conjugate_flag)
% Same signature as Primal Dual Optimization/SAD_denoise.m:1.
%   G = SAD_denoise([], g, lambda, tau)   operator struct for chambolle_pock (engine path)
%   numeric u: the original function, kept as SAD_denoise_matlab.m
if isempty(u)
    res1 = struct('type', 'SAD_denoise', 'g', g(:), 'lambda', lambda, 'tau', tau);
    return
end
if nargin < 5, conjugate_flag = false; end
if nargout == 3
    [res1, res2, res3] = SAD_denoise_matlab(u, g, lambda, tau, conjugate_flag);
else
    [res1, res2] = SAD_denoise_matlab(u, g, lambda, tau, conjugate_flag);
end
end
