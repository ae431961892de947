function cp_print_log(Fname, Gname, theta, tau, sigma, nx, ny, maxIter, tol, primal_history, dual_history, stopped_by_tol)
% The console log of Primal Dual Optimization/chambolle_pock.m:109-131 (header, row 0), :182-191 (every 10th iteration)
% and :196-203 (stopping criterion), printed after the solve from the histories the engine returns: column 1 = p / q,
% the last two columns = the constraint measures (F, G) / (F*, G*) (:225-228).
fprintf('\nCHAMBOLLE POCK PRIMAL DUAL OPTIMIZATION SCHEME\n');
fprintf('\n\tPRIMAL PROBLEM\tp(x) = F(Kx) + G(x)\t\t-> min!\n');
fprintf('\tDUAL PROBLEM\tq(y) = -G*(-K*y) - F*(y)\t-> max!\n');
fprintf('\nFOR\n\n\tF = %s\n\tG = %s\n', Fname, Gname);
fprintf('\tGAP(x,y) = |p(x)-q(y)| / |q(y)|\n');
fprintf('\nWITH PARAMETERS\n');
fprintf('\n\tEXTRAGRADIENT STEP SIZE\t\tTHETA\t= %.3f', theta);
fprintf('\n\tPRIMAL STEP SIZE\t\tTAU\t= %.3f', tau);
fprintf('\n\tDUAL STEP SIZE\t\t\tSIGMA\t= %.3f\n', sigma);
fprintf('\n\tNUMBER OF PRIMAL VARIABLES\t %d', nx);
fprintf('\n\tNUMBER OF DUAL VARIABLES\t %d\n', ny);
fprintf('\n\tMAX NUMBER OF ITERATIONS\t\t %d', maxIter);
fprintf('\n\tTOLERANCE FOR GAP & INFEASIBILITIES\t %.1e\n', tol);
fprintf('\ni\tp(x_i)\t\tq(y_i)\t\tGAP(x_i,y_i)\tINFEASIBILITIES\n');
fprintf([repmat('-', [1, 116]), '\n']);
rows = size(primal_history, 1);
for r = [1, 11 : 10 : rows]
    p = primal_history(r, 1);
    q = dual_history(r, 1);
    fprintf('%d\t%+.2e\t%+.2e\t%.3e', r - 1, p, q, abs((p - q) / q));
    fprintf('\tF: %.2e', primal_history(r, end - 1));
    fprintf('\tG: %.2e', primal_history(r, end));
    fprintf('\tF*: %.2e', dual_history(r, end - 1));
    fprintf('\tG*: %.2e', dual_history(r, end));
    fprintf('\n');
end
fprintf('\nSTOPPING AT CRITERION:\n\n\t');
if stopped_by_tol
    fprintf('TOLERANCE FOR GAP & INFEASIBILITIES REACHED\n\n');
else
    fprintf('MAX NUMBER OF ITERATIONS REACHED\n\n');
end
end
