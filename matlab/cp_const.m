function c = cp_const()
% numeric ids of include/cpengine.h (enum cp_model, cp_bc, cp_flag, CP_STOP_*)
c.MODEL_TV_L1_DENOISE = 1; c.MODEL_TV_L1_REG = 2; c.MODEL_TV_L2_REG = 3;
c.MODEL_MF_NN_TV = 4; c.MODEL_MF_NN_CURV = 5; c.MODEL_VAR_TV = 6; c.MODEL_VAR_CURV = 7;
c.BC_NEUMANN = 0; c.BC_LINEAR = 1; c.BC_DIRICHLET = 2;
c.FLAG_VAR_SSD_SWAP = 1; c.FLAG_FP32 = 2; c.FLAG_MFNN_NORM21 = 4;
c.STOP_MAXITER = 0; c.STOP_TOLERANCE = 1;
end
