// Argument block of the rollout kernel (shared by the host launcher and the kernel TUs).
#pragma once
#include "common.h"

struct RolloutArgs {
    const float *wstream, *bstream;
    unsigned wbytes;                  // size of the whole weight stream buffer (all members)
    unsigned wmember_b;               // bytes per member
    unsigned w_l0_b, w_lh_b, w_lo_b;  // layer stream sizes (bytes): L0, hidden, OUT
    const unsigned short* xw;         // split-f16 fragment stream (xdl kernel)
    unsigned xw_bytes, xw_member_b, xw_wave_b[8];
    const float* xb;                  // its bias tiles
    size_t xb_member;
    size_t bmember;                   // bias floats per member
    size_t b_l0, b_lh;                // bias tile sizes (floats): L0/hidden, (OUT follows)
    const float *obs, *obs_rows, *ctx_vec, *actions, *eps;
    const float *obs_mean, *obs_std, *act_mean, *act_std, *delta_mean, *delta_std, *maxlv, *minlv;
    float *returns_rows, *traj;
    int m, n_local, n_global, cand_offset, E, p, PE, H, NH, it, quirks, deterministic, norm_actions;
    uint32_t seed, call;
    int wgs_per_member, rows_per_member;
    int tile0, tile_count;            // xdl kernel: row tiles [tile0, tile0 + tile_count) of every member in this launch
    int bias_lds;                     // xdl kernel: bias tiles staged in LDS
    int dry_run;                      // launcher: validate the geometry (LDS, instantiation) without launching
    unsigned long long* tbuf;   // CADM_PHASE_TIMING builds only
    const unsigned short* xw1;        // wave-tile kernel (rollout_wt.h): the fragment stream in its one-wave order
    unsigned xw1_member_b;
    int wt_waves;                     // wave-tile kernel: row tiles (= active waves) per workgroup and round
};
