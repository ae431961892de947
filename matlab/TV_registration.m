function [res1, res2, res3] = TV_registration(v, sigma, conjugate_flag)
% Same signature as Primal Dual Optimization/TV_registration.m:1.
%   F = TV_registration([], sigma)            operator struct for chambolle_pock (engine path); replaces the closure
%                                     F = @(v, c_flag) TV_registration(v, sigma, c_flag)
%   [..] = TV_registration(v, sigma, flag)    numeric v: the reference's own TV_registration.m (cp_call_shadowed)
if isempty(v)
    res1 = struct('type', 'TV_registration', 'sigma', sigma);
    return
end
if nargin < 3, conjugate_flag = false; end
if nargout == 3
    [res1, res2, res3] = cp_call_shadowed('TV_registration', v, sigma, conjugate_flag);
else
    [res1, res2] = cp_call_shadowed('TV_registration', v, sigma, conjugate_flag);
end
end
