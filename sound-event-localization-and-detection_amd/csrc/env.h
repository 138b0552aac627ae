// Environment switches of libseld_hip.so, read ONCE (first use) into a validated table; seld_env_reload() re-reads
// them (the tests switch kernel generations inside one process).
//
// Selection switches -- every setting computes the same results, they only choose which kernel generation runs
// (the test suite uses them to cover every generation):
//   SELD_CONV_CFG=ct,pt   force the convolution tile (one of the candidates of pick_cfg, anything else is ignored)
//   SELD_CONV_NOVEC / SELD_CONV_NOFAST / SELD_CONV_NO_SMALLK / SELD_NO_FWD_PAIR / SELD_CONV_PAIR / SELD_CONV_NO_HCQ / SELD_HCQ_NO_FIRST / SELD_HCQ_NO_POOL / SELD_HCQ_WGRAD_DQ / SELD_HCQ_WGRAD_ROW
//   SELD_WGRAD_NOROW / SELD_WGRAD_SLOW / SELD_WGRAD_CFG=0..5 / SELD_WGRAD_WGS=n / SELD_SMALLK_WGS=n
//   SELD_MHA_NO_MFMA
//   SELD_DETERMINISTIC    run-to-run reproducible results: reductions that are normally split over workgroups and folded
//                         with float atomics (BatchNorm statistics, weight-gradient splits, bias / loss sums) run as ONE
//                         ordered chain per output element -- same values up to summation order, slower
// Timing-experiment switches that switch parts of a kernel OFF and therefore give WRONG results exist only in
// builds compiled with -DSELD_TUNING (never the shipped library):
//   SELD_VEC_DBG, SELD_WGRAD_DBG, SELD_SMALLK_DBG, SELD_SMALLK_NW
#pragma once

namespace seld {

struct SeldEnv {
    int conv_cfg_ct = 0, conv_cfg_pt = 0;          // 0 = not forced
    bool conv_novec = false, conv_nofast = false, conv_no_smallk = false, no_fwd_pair = false, conv_pair = false;
    bool hcq_wgrad_dq = false;                      // SELD_HCQ_WGRAD_DQ: fast-product weight gradient for the dual quaternion too
    bool conv_no_hcq = false;                       // SELD_CONV_NO_HCQ: 16/48-product kernels instead of hcq_conv.hip
    bool hcq_wgrad_row = false;                     // SELD_HCQ_WGRAD_ROW: dual-quaternion weight gradients on the 24-product row kernel
    bool hcq_no_pool = false;                       // SELD_HCQ_NO_POOL: first stage without the pooling convolution kernel
    bool stft_radix2 = false;                       // SELD_STFT_RADIX2: nperseg 512 on the general radix-2 kernel too
    bool hcq_no_first = false;                      // SELD_HCQ_NO_FIRST: first layers without the row-walking kernel
    bool deterministic = false;                     // SELD_DETERMINISTIC: every reduction in a fixed order (no multi-contributor float atomics)
    bool wgrad_norow = false, wgrad_slow = false, mha_no_mfma = false;
    int wgrad_cfg = -1;                             // -1 = not forced, else 0..5
    long long wgrad_wgs = 0, smallk_wgs = 0;        // 0 = default
    int vec_dbg = 0, wgrad_dbg = 0, smallk_dbg = 0, smallk_nw = 4;   // SELD_TUNING builds only; otherwise the defaults
};

const SeldEnv& env();      // abi.hip

}  // namespace seld
