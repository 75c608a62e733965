// back_pass_mfma_lims.hip — the control-limited (boxQP) instantiation of the n = 64, m = 8 matrix-core backward pass
#include "back_pass_mfma_kernel.h"

int ddp_bpm_launch_lims(ddp_handle h, const BPMArgs &a) { return ddp_bpm_launch<true>(h, a); }
