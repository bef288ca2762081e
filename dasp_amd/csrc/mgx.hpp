// mgx.hpp -- the direct ("push") exchange of the multi-GPU step: kernel arguments and launchers (mgx.hip), used by multigpu.cpp
#pragma once
#include <cstddef>

namespace dasp {

struct MgPushDst {
    void *data;                      // where this rank's slice goes in the destination's gather buffer (local or peer-mapped)
    unsigned long long *flag;        // the destination's arrived[this rank] word
};

struct MgPushArgs {
    const void *src;                 // this rank's padded slice of y
    size_t bytes;                    // its size (a multiple of 16)
    const MgPushDst *dst;            // device table, one entry per destination
    int n_dst;
    unsigned *count;                 // device, one zero-initialised counter per destination (never reset: counts modulo wgs)
    int wgs;                         // workgroups of the launch: each moves one part of the slice to every destination
    unsigned long long seq;          // what the flags are set to
    const unsigned long long *ready; // fused step: wait for *ready >= ready_need first (ready_need 0: no wait)
    unsigned long long ready_need;
    long long timeout;               // 100 MHz ticks
    int *err;                        // sticky error word of the step (2: the wait for the product timed out)
    long long delay_ticks;           // TEST HOOK (loopback timing): the flags go up this long after the stores (0 in every real exchange)
};

int launch_mg_push(const MgPushArgs &a, void *stream);
// one wave waits for arrived[0..world) >= seq, then (gathered != nullptr) *gathered = step
int launch_mg_arrived(const void *arrived, int world, unsigned long long seq, void *gathered, unsigned long long step, long long timeout_ticks,
                      void *err, void *stream, int skip = -1);      // skip: a rank whose flag is not waited for (the one-stream step: this rank itself)

}  // namespace dasp
