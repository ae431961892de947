function LM_acc = landmark_accuracy(LM)
% LM_acc = landmark_accuracy(LM) -- drop-in for NNE/landmark_accuracy.m (cp_landmark_accuracy through the MEX gateway):
% LM is a cell array of k numLM x 2 landmark sets; LM_acc(l) = mean distance of landmark l to its mean position.
LM_acc = cp_mex('landmark_accuracy', cat(3, LM{:}));
end
