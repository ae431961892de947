function LM_transformed = landmark_transform(LM, u, omega)
% LM_transformed = landmark_transform(LM, u, omega) -- drop-in for NNE/landmark_transform.m on the MI355X engine
% (cp_landmark_transform through the MEX gateway): nearest deformed grid point, then the fixed-point iteration
% x <- x + (y - F(x)) with the reference's termination expression and its warning + fallback to the initial guess.
%   LM ~ numLM x 2, u ~ m x n x 2, omega ~ 1 x 4
[LM_transformed, failed] = cp_mex('landmark_transform', LM, u, omega);
for i = 1 : nnz(failed)
    warning('Fixed-point iteration failed! Returning initial guess.');
end
end
