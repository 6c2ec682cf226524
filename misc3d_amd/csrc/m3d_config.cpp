// m3d_config.cpp -- environment -> m3d_config, once (see m3d_config.hpp).
#include "m3d_config.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "m3d_driver.hpp"

namespace m3d {
namespace {
m3d_config g_cfg;
std::once_flag g_once;
std::mutex g_cfg_mu;                 // guards g_cfg against m3d_set_config
thread_local m3d_config t_cfg;       // the calling thread's view (pinned: the snapshot; else refreshed by every config())
thread_local int t_cfg_pins = 0;

long env_long(const char* name, long def) {
    const char* e = std::getenv(name);
    if (!e || !*e) return def;
    char* end = nullptr;
    const long v = std::strtol(e, &end, 10);
    return end == e ? def : v;
}
bool env_is(const char* name, char first) {
    const char* e = std::getenv(name);
    return e && e[0] == first;
}
void sanitize(m3d_config& c) {
    if (c.lead_hypotheses < 64 || c.lead_hypotheses % 64) c.lead_hypotheses = 128;
    if (c.score_groups_per_block < 1 || c.score_groups_per_block > 64) c.score_groups_per_block = 8;
    if (c.score_min_workgroups < 1) c.score_min_workgroups = 8192;
    if (c.dense_workgroups < 1) c.dense_workgroups = 8192;
    if (c.pool_limit_mb < 0) c.pool_limit_mb = 0;
    if (c.score_mfma_groups < 1 || c.score_mfma_groups > 64) c.score_mfma_groups = 64;
    if (c.score_waves4_groups < 1 || c.score_waves4_groups > 64) c.score_waves4_groups = 64;
    if (c.score_phases < -1 || c.score_phases == 1 || c.score_phases > 3) c.score_phases = -1;
    if (c.plane_bound < 0 || c.plane_bound > 2) c.plane_bound = 1;
    if (c.cull_fp32 < 0 || c.cull_fp32 > 2) c.cull_fp32 = 1;
    if (!kExperimentalBuild) c.score_mfma = c.score_waves4 = c.compact_one_pass = 0;   // (not compiled in: m3d_kernels.hpp)
    if (c.lanes < 1 || c.lanes > 8) c.lanes = 4;
    if (c.wait_spin_us < 0) c.wait_spin_us = 500;
    c.prestream = c.prestream != 0;
    c.chunk_cap = (int32_t)std::min<long>(std::max<long>(((long)c.chunk_cap + 63) / 64 * 64, 1024), 262144);
    c.first_chunk = c.first_chunk <= 0 ? 0 : (int32_t)std::min<long>(((long)c.first_chunk + 63) / 64 * 64, 1 << 20);
    if (c.reg_cells_per_radius < 1 || c.reg_cells_per_radius > 16) c.reg_cells_per_radius = 4;
    if (c.match_pipeline < 0 || c.match_pipeline > 2) c.match_pipeline = 1;
    if (c.reg_cache < 0 || c.reg_cache > 2) c.reg_cache = 1;
    if (c.device_aliases < 0) c.device_aliases = 0;
    if (c.device_aliases > 16) c.device_aliases = 16;
    if (c.lanes_eager < 1 || c.lanes_eager > 8) c.lanes_eager = 2;
}
void load_env() {
    std::memset(&g_cfg, 0, sizeof(g_cfg));
    g_cfg.dense_scoring = env_is("M3D_DENSE", '1');
    g_cfg.speculative_refine = !env_is("M3D_SPEC", '0');
    g_cfg.lead_hypotheses = (int32_t)env_long("M3D_LEAD", 128);        // sweep on C2: 64 and 128 equal, 256 +2.5 %, 512 +4 %
    g_cfg.score_groups_per_block = (int32_t)env_long("M3D_GPB", 8);
    g_cfg.score_min_workgroups = (int32_t)env_long("M3D_SCORE_MIN_WGS", 8192);   // C5 (1000 hypotheses per round on ~1 M points): 4096 32.7, 8192 31.8, 16384 34.0 ms; C2 flat
    g_cfg.dense_workgroups = (int32_t)env_long("M3D_SCORE_WGS", 8192);  // sweep on MI355X: 2048 +5 %, 4096 +1.5 %, 8192..32768 flat
    g_cfg.reg_neighbour_lists = !env_is("M3D_REG_NL", '0');
    g_cfg.reg_prune = !env_is("M3D_REG_PRUNE", '0');
    g_cfg.match_brute = env_is("M3D_MATCH_BRUTE", '1');
    g_cfg.match_fp32_screen = env_is("M3D_MATCH_SCREEN", 'f');
    g_cfg.pool_limit_mb = (int32_t)env_long("M3D_POOL_MB", 4096);
    g_cfg.kernel_timing = env_is("M3D_KERNEL_TIMING", '1');
    g_cfg.reg_sorted_lists = !env_is("M3D_REG_SORTED", '0');
    g_cfg.score_fp32_screen = !env_is("M3D_SCORE_SCREEN", '0');
    g_cfg.cull_fp32 = env_is("M3D_CULL_FP32", '0') ? 0 : (env_is("M3D_CULL_FP32", '2') ? 2 : 1);
    g_cfg.reg_fp32_screen = !env_is("M3D_REG_SCREEN", '0');
    g_cfg.sorted_tombstones = !env_is("M3D_TOMBSTONES", '0');
    g_cfg.score_mfma = env_is("M3D_SCORE_MFMA", '1');
    g_cfg.score_mfma_groups = (int32_t)env_long("M3D_MFMA_GPB", 64);
    g_cfg.score_waves4 = env_is("M3D_SCORE_WAVES4", '1');   // C2 0.1059 ms against 0.1020 (one-wave workgroups), C3 equal: off
    g_cfg.score_waves4_groups = (int32_t)env_long("M3D_WAVES4_GPB", 64);
    g_cfg.score_phases = (int32_t)env_long("M3D_SCORE_PHASES", -1);
    g_cfg.compact_one_pass = env_is("M3D_COMPACT_ONE_PASS", '1');   // C5 rounds 17.4 against 15.3 ms, C2 step +8 us: off (profiles/r04_compact_one_pass.txt)
    g_cfg.plane_bound = (int32_t)env_long("M3D_PLANE_BOUND", 1);
    g_cfg.lanes = (int32_t)env_long("M3D_LANES", 4);
    g_cfg.wait_spin_us = (int32_t)env_long("M3D_SPIN_US", 500);
    g_cfg.prestream = !env_is("M3D_PRESTREAM", '0');
    g_cfg.chunk_cap = (int32_t)env_long("M3D_CHUNK_CAP", 24576);
    g_cfg.first_chunk = (int32_t)env_long("M3D_FIRST_CHUNK", 2048);
    g_cfg.reg_cells_per_radius = (int32_t)env_long("M3D_REG_K", 4);
    g_cfg.match_pipeline = (int32_t)env_long("M3D_MATCH_PIPELINE", 1);
    g_cfg.reg_cache = (int32_t)env_long("M3D_REG_CACHE", 1);
    g_cfg.device_aliases = (int32_t)env_long("M3D_DEVICE_ALIASES", 0);
    g_cfg.lanes_eager = (int32_t)env_long("M3D_LANES_EAGER", 2);
    sanitize(g_cfg);
}
}  // namespace

const m3d_config& config() {
    std::call_once(g_once, load_env);
    if (t_cfg_pins > 0) return t_cfg;
    std::lock_guard<std::mutex> lock(g_cfg_mu);
    t_cfg = g_cfg;
    return t_cfg;
}
void config_pin() {
    std::call_once(g_once, load_env);
    if (t_cfg_pins++ == 0) {
        std::lock_guard<std::mutex> lock(g_cfg_mu);
        t_cfg = g_cfg;
    }
}
void config_unpin() {
    if (t_cfg_pins > 0) --t_cfg_pins;
}
void config_store(const m3d_config& c) {
    std::call_once(g_once, load_env);
    m3d_config n = c;
    sanitize(n);
    std::lock_guard<std::mutex> lock(g_cfg_mu);
    g_cfg = n;
}
}  // namespace m3d

extern "C" {
void m3d_get_config(m3d_config* out) {
    if (out) *out = m3d::config();
}
int m3d_set_config(const m3d_config* in) {
    if (!in) return m3d::fail(M3D_ERR_INVALID_ARG, "invalid argument");
    // the three switches of round 4's refuted variants select kernels a product build does not contain (m3d_kernels.hpp): asking
    // for one is an error, not a silent no-op; their slots stay for the layout of the fields behind them
    if (!m3d::kExperimentalBuild && (in->score_mfma || in->score_waves4 || in->compact_one_pass))
        return m3d::fail(M3D_ERR_INVALID_ARG,
                         "score_mfma / score_waves4 / compact_one_pass: these kernels are compiled with -DM3D_EXPERIMENTAL only "
                         "(m3d_bench_experimental() == 0)");
    m3d::config_store(*in);
    return M3D_OK;
}
}
