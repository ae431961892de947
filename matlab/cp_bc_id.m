function id = cp_bc_id(bc)
% boundary condition string of optPara.bc -> CP_BC_*; unknown strings fail in the library with the message id
% cpengine:badBC ('Unknown boundary condition!', finite_difference_operator.m:35)
switch lower(bc)
    case 'neumann',   id = 0;
    case 'linear',    id = 1;
    case 'dirichlet', id = 2;
    otherwise,        id = -1;     % rejected by cp_create with cpengine:badBC
end
end
