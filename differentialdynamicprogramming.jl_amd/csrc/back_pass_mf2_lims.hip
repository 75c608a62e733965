// back_pass_mf2_lims.hip — the control-limited (boxQP) instantiations of the large-state matrix-core backward pass
#include "back_pass_mf2_kernel.h"

int ddp_bpm2_launch_lims(ddp_handle h, const BPM2Args &a, int nt) { return nt == 4 ? mf2::launch<4, true>(h, a) : mf2::launch<3, true>(h, a); }
