function K = cp_gradient_operator(m, n, h, ncomp)
% K = kron(I_ncomp, [Gx; Gy]) with Neumann rows (test_TV_L1_registration.m:67-72, test_TV_L1_denoising.m:43-49) as a
% struct the engine recognises; the matrix itself is never formed (the engine applies it as a stencil).
K = struct('type', 'gradient', 'm', m, 'n', n, 'h', h(:)', 'ncomp', ncomp);
end
