// debug_sched.cpp — DEBUG ONLY: prints the statistics the counting kernel variants tally (RTTNW_DEBUG_SCHED=1 with
// rttnw_params.collect_counters 1-3): wave clock per phase, lockstep iterations and the lanes they served, walk-length
// histograms.  Nothing here runs in a render without that environment variable; kept out of the launch path's source
// (render_tiles.hpp).  The meaning of DeviceCounters::dbg[] per kernel is documented where it is tallied (trace_tally.hpp,
// trace_kernels.hpp).  profiles/r0N/phases_*.txt are this output (since round 5 of the
// kernel's asynchronous form: a PHASE = walks started, trips until few walks are unfinished, the finished lanes shaded).
#include "render_common.hpp"

namespace rt {

void debug_print_sched(const DeviceCounters& hc, bool plain, uint32_t profile, uint64_t samples) {
    if (plain) {
        const double tot = double(hc.dbg[0] + hc.dbg[1] + hc.dbg[2] + hc.dbg[3]);
        fprintf(stderr, "[plain] wave clock: hand-out %.1f%%  begin %.1f%%  walk %.1f%%  shade %.1f%% (media + hit record %.1f%%, material %.1f%%)\n", 100 * hc.dbg[0] / tot,
                100 * hc.dbg[1] / tot, 100 * hc.dbg[2] / tot, 100 * hc.dbg[3] / tot, 100 * hc.dbg[15] / tot, 100 * (hc.dbg[3] - hc.dbg[15]) / tot);
        fprintf(stderr, "[plain] walk: %.1f trips/phase (%.1f with node lanes, %.1f with leaf lanes); lane steps served per trip %.1f\n",
                double(hc.dbg[4]) / hc.dbg[9], double(hc.dbg[7]) / hc.dbg[9], double(hc.dbg[8]) / hc.dbg[9],
                double(hc.dbg[5] + hc.dbg[6]) / hc.dbg[4]);
        fprintf(stderr, "[plain] walk clock: node steps %.1f%%, leaf steps %.1f%% of the walk\n", 100.0 * hc.dbg[13] / hc.dbg[2], 100.0 * hc.dbg[14] / hc.dbg[2]);
        for (uint32_t m = 1; m < 64 && profile == 2u; ++m) // (collect_counters = 2 only: level 3 keeps its histograms in the same words)
            if (hc.dbg[80 + m] * 200 > hc.dbg[8])
                fprintf(stderr, "[plain]   leaf iterations serving {%s%s%s%s%s%s}: %.1f%% of them, %.1f%% of the leaf clock, %.0f clocks each\n", m & 1 ? "sphere " : "",
                        m & 2 ? "moving " : "", m & 4 ? "rect " : "", m & 8 ? "box " : "", m & 16 ? "instance " : "", m & 32 ? "empty " : "",
                        100.0 * hc.dbg[80 + m] / hc.dbg[8], 100.0 * hc.dbg[16 + m] / hc.dbg[14], double(hc.dbg[16 + m]) / hc.dbg[80 + m]);
        fprintf(stderr, "[plain]   node iterations: %.0f clocks each\n", double(hc.dbg[13]) / hc.dbg[7]);
        if (profile == 3u) { // collect_counters = 3: distribution of walk lengths, in trips
            fprintf(stderr, "[plain] trips per walk (lanes):");
            for (int k = 0; k < 64; ++k) fprintf(stderr, " %llu", hc.dbg[16 + k]);
            fprintf(stderr, "\n[plain] walks / mean trips by result (miss, sphere, moving, rect, box, -, in instance):");
            for (int k = 0; k < 7; ++k) fprintf(stderr, " %llu / %.1f", hc.dbg[152 + k], hc.dbg[152 + k] ? double(hc.dbg[144 + k]) / hc.dbg[152 + k] : 0.0);
            fprintf(stderr, "\n[plain] trips of a phase (waves):");
            for (int k = 0; k < 64; ++k) fprintf(stderr, " %llu", hc.dbg[80 + k]);
            fprintf(stderr, "\n");
        }
        fprintf(stderr, "[plain] node lane-steps per trip with node lanes %.1f (a trip has 2 or 3 node steps), leaf lanes per leaf step %.1f; phases/sample %.2f, lanes shaded per phase %.1f; begin in %.0f%% of phases, %.1f lanes each\n",
                double(hc.dbg[5]) / hc.dbg[7], double(hc.dbg[6]) / hc.dbg[8], double(hc.dbg[9]) * 64 / samples,
                double(hc.dbg[10]) / hc.dbg[9], 100.0 * hc.dbg[11] / hc.dbg[9], hc.dbg[11] ? double(hc.dbg[12]) / hc.dbg[11] : 0.0);
    } else {
        const double w64 = double(samples) / 64.0;
        fprintf(stderr, "[decoupled] bursts/64smp %.1f  shades/64smp %.2f (lanes %.1f)  refills/64smp %.1f (lanes %.1f)\n", hc.dbg[8] / w64, hc.dbg[9] / w64,
                hc.dbg[9] ? double(hc.dbg[10]) / hc.dbg[9] : 0.0, hc.dbg[11] / w64, hc.dbg[11] ? double(hc.dbg[12]) / hc.dbg[11] : 0.0);
    }
}

} // namespace rt
