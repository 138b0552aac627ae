// ABI bookkeeping entry points of libseld_hip.so.
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include "common.h"
#include "env.h"

namespace seld {
thread_local int g_last_hip_error = 0;

static SeldEnv g_env;
static bool g_env_loaded = false;
static std::mutex g_env_mutex;

static bool flag(const char* name) { const char* e = getenv(name); return e && *e; }
static long long bounded(const char* name, long long lo, long long hi, long long dflt) {
    const char* e = getenv(name);
    if (!e || !*e) return dflt;
    char* end = nullptr;
    const long long v = strtoll(e, &end, 10);
    return (end == e || v < lo || v > hi) ? dflt : v;       // out-of-range or malformed values are ignored
}

static void load_env_locked() {
    SeldEnv e;
    if (const char* c = getenv("SELD_CONV_CFG")) {
        int a = 0, b = 0;
        if (sscanf(c, "%d,%d", &a, &b) == 2 && a >= 1 && a <= 12 && b >= 1 && b <= 4) { e.conv_cfg_ct = a; e.conv_cfg_pt = b; }
    }
    e.conv_novec = flag("SELD_CONV_NOVEC");
    e.conv_nofast = flag("SELD_CONV_NOFAST");
    e.conv_no_smallk = flag("SELD_CONV_NO_SMALLK");
    e.no_fwd_pair = flag("SELD_NO_FWD_PAIR");
    e.conv_pair = flag("SELD_CONV_PAIR");
    e.conv_no_hcq = flag("SELD_CONV_NO_HCQ");
    e.hcq_wgrad_dq = flag("SELD_HCQ_WGRAD_DQ");
    e.stft_radix2 = flag("SELD_STFT_RADIX2");
    e.hcq_no_first = flag("SELD_HCQ_NO_FIRST");
    e.hcq_no_pool = flag("SELD_HCQ_NO_POOL");
    e.hcq_wgrad_row = flag("SELD_HCQ_WGRAD_ROW");
    e.deterministic = flag("SELD_DETERMINISTIC");
    e.wgrad_norow = flag("SELD_WGRAD_NOROW");
    e.wgrad_slow = flag("SELD_WGRAD_SLOW");
    e.mha_no_mfma = flag("SELD_MHA_NO_MFMA");
    e.wgrad_cfg = (int)bounded("SELD_WGRAD_CFG", 0, 5, -1);
    e.wgrad_wgs = bounded("SELD_WGRAD_WGS", 1, 1 << 20, 0);
    e.smallk_wgs = bounded("SELD_SMALLK_WGS", 1, 1 << 20, 0);
#ifdef SELD_TUNING
    e.vec_dbg = (int)bounded("SELD_VEC_DBG", 0, 63, 0);
    e.wgrad_dbg = (int)bounded("SELD_WGRAD_DBG", 0, 3, 0);
    e.smallk_dbg = (int)bounded("SELD_SMALLK_DBG", 0, 1023, 0);
    e.smallk_nw = bounded("SELD_SMALLK_NW", 4, 8, 4) == 8 ? 8 : 4;
#endif
    g_env = e;
    g_env_loaded = true;
}

const SeldEnv& env() {
    if (!g_env_loaded) {
        std::lock_guard<std::mutex> lk(g_env_mutex);
        if (!g_env_loaded) load_env_locked();
    }
    return g_env;
}
}  // namespace seld

/* Re-read the SELD_* environment switches (they are read once, at first use).  Not thread safe against
 * concurrent launches: call it between launches (the tests do, after changing the environment). */
extern "C" int seld_env_reload(void) {
    std::lock_guard<std::mutex> lk(seld::g_env_mutex);
    seld::load_env_locked();
    return SELD_OK;
}
/* 1 if this library was built with -DSELD_TUNING (wrong-result timing switches compiled in), else 0 */
extern "C" int seld_tuning_build(void) {
#ifdef SELD_TUNING
    return 1;
#else
    return 0;
#endif
}

extern "C" int seld_abi_version(void) { return 2; }
extern "C" const char* seld_build_arch(void) { return "gfx950"; }
extern "C" int seld_last_hip_error(void) { return seld::g_last_hip_error; }
