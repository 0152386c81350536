// Launch-timing registry behind prof.h (one definition for the whole library) and its C ABI (include/tclight_hip.h).
#include "common.h"
#include "prof.h"

TclProfClass g_tcl_prof[TCL_PROF_NCLS];

extern "C" {

// Switch event timing on for the kernel classes in `mask` (bit 0: GEMM / implicit-conv family, work = 2 M N K FLOP per call; bit 1: VidToMe
// match calls, work = 2 n_src n_dst C B FLOP per call) and reset their accumulators.
int tcl_prof_begin(int mask) {
    for (int c = 0; c < TCL_PROF_NCLS; ++c) {
        TclProfClass& p = g_tcl_prof[c];
        p.drain(true);
        p.on = (mask >> c) & 1; p.ms = 0.0; p.work = 0.0; p.launches = 0;
    }
    return TCL_OK;
}
// -> summed event time (ms), summed algorithmic work and call count of class `cls` since tcl_prof_begin; synchronises its events, switches it off.
int tcl_prof_end(int cls, double* total_ms, double* total_work, long* launches) {
    TCL_CHECK_ARG(cls >= 0 && cls < TCL_PROF_NCLS && total_ms && total_work && launches);
    TclProfClass& p = g_tcl_prof[cls];
    p.drain(true);
    *total_ms = p.ms; *total_work = p.work; *launches = p.launches;
    p.on = false;
    return TCL_OK;
}

}  // extern "C"
