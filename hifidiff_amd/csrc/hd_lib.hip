// hd_lib.hip — host side of libhifidiff_hip.so: context, weight ingest (fold + pack), the launch
// program of one denoiser evaluation, the once-per-batch conditioning prologue, and the graph-replayed
// reverse-diffusion loop.  C-ABI in include/hifidiff_hip.h.  gfx950 only.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/hifidiff_hip.h"
#include "hd_chain.hpp"
#include "hd_conv.hpp"
#include "hd_cr.hpp"
#include "hd_dispatch.hpp"
#include "hd_gemm.hpp"
#include "hd_kernels.hpp"
#include "hd_vae.hpp"
#include "hd_stage_api.hpp"

using namespace hd;

namespace {

constexpr int WIDTH = 128;
constexpr int FILM_IN = 256;                  // SimpleGate(512) -> 256 (conditional_naf.py:19)
constexpr size_t HOST_MIRROR_MAX = 1 << 16;   // small tensors keep a host copy (BN folding etc.)

struct RawTensor {
    std::vector<int64_t> shape;
    size_t numel = 0;
    float* dev = nullptr;
    std::vector<float> host;                  // filled when numel <= HOST_MIRROR_MAX
};

struct PackedW {
    uint4* w = nullptr;
    int N = 0, K = 0, Kp = 0, nt_total = 0;   // K = ntaps*Cin_pad
    const float* bias = nullptr;              // device fp32 (may be folded)
};

struct BlockW {
    std::string name;
    int C = 0, film_off = -1;
    PackedW conv1, conv3, sca, conv4, conv5;
    const float *dw_w = nullptr, *dw_b = nullptr, *beta = nullptr, *gamma = nullptr;
    const float* dw_wT = nullptr;             // conv2.weight tap-major [9][2C] (fused conv1 epilogue)
};

struct HcaW {
    int C = 0;
    PackedW mlp0, mlp2, sp0, fused;
    const float* sp3_w = nullptr;             // folded [C/2]
    float sp3_b = 0.f;
    bool centre_only = false;
};

struct ResConv { PackedW w; int cin, cout, k, stride, pad; };
struct ResBlock { ResConv c1, c2, c3, ds; bool has_ds = false; };


struct Op {
    std::string name;
    std::function<hipError_t(hipStream_t)> run;
    std::shared_ptr<GemmP> gemm;     // launch parameters of a GEMM op (patched by link_prefetch), else null
    bool skinny_affine = false;      // runs in the skinny kernel with the XCD-affine tile map
    const void* out = nullptr;       // output buffer of the launch (introspection only)
    size_t out_elems = 0;
    int out_bf16 = 0;
};

struct Level { int C, H, M; float *X, *Y, *T1, *pooled, *S; unsigned short *G, *Xb, *Yb, *Xg, *pooled16; float2 *sx, *sy; };

}  // namespace

// One independently scheduled sub-batch: faces never interact inside the loop, so the batch is cut into
// chains whose launch sequences run concurrently (forked branches of the captured graph) and overlap
// each other's latency-bound phases.
struct Chain {
    int B = 0, face0 = 0, index = 0;
    Level lv[5];
    float *lat = nullptr, *eps = nullptr, *prior[5] = {}, *gate_c[5] = {}, *gate_s[5] = {}, *idc_term = nullptr;
    float *id_emb = nullptr, *pool_tmp = nullptr, *mlp_tmp = nullptr, *sp_tmp = nullptr;
    unsigned short* res_buf[4] = {};
    uint4* face8 = nullptr;
    std::vector<Op> program;                 // one denoiser evaluation (lat -> eps) of this chain's faces
    std::vector<Op> prep_program;            // the most recent conditioning prologue
    StepState* step_state = nullptr;         // device-resident loop state of this chain
    float* film_cur = nullptr;               // FiLM row of the step being evaluated (sampling loop; see sched_step_kernel)
    hipStream_t stream = nullptr;            // the chain's own queue for the sampling loop
    hipEvent_t done = nullptr;
    hipGraphExec_t graph_exec = nullptr;     // program + scheduler update of this chain, replayed per step
    hipGraphExec_t graph_multi = nullptr;    // kGraphSteps consecutive steps in one graph (the step index lives in device memory)
};

static std::string g_create_error;
constexpr int kGraphSteps = 10;              // diffusion steps per captured graph in hd_sample (plus a one-step graph for the remainder)

// Everything whose size depends on the batch: buffers, launch programs, captured graphs.  One context serves any batch
// size (the reference's val loop has a ragged last batch: DataLoader without drop_last, test_refiner.py:160): the
// workspace of the batch in use lives in hd_ctx itself, workspaces of other recent batch sizes are parked here
// (packed weights are shared and never touched).
struct VaeWs { float *X = nullptr, *T = nullptr, *S = nullptr, *mom = nullptr, *Q = nullptr, *K = nullptr, *V = nullptr, *resz = nullptr, *out3 = nullptr;
               unsigned short *H = nullptr, *H2 = nullptr, *Xb = nullptr, *U = nullptr; uint4* in8 = nullptr; double* part = nullptr; int B = 0, R = 0; };

struct SavedWs {
    VaeWs vws;
    std::vector<Op> vae_enc_prog, vae_dec_prog;
    int B = 0;
    uint64_t stamp = 0;
    std::vector<Chain> chains;
    float *lat = nullptr, *eps = nullptr;
    std::vector<void*> allocs;
    std::map<std::string, std::pair<void*, std::pair<size_t, int>>> dbg;
    bool graphs_valid = false;
    const float* graph_film = nullptr;
    int graph_B = 0;
    // CoarseRestoration contexts
    std::vector<Op> cr_program;
    const float* cr_in = nullptr; float* cr_out = nullptr;
    float* cr_skip[5] = {};
    float *cr_loc1 = nullptr, *cr_loc2 = nullptr, *cr_theta = nullptr;
};

struct hd_ctx {
    int L = 16, device = 0, S = 1;            // S = L/16
    bool conditional = true;                  // false: the unconditional Denoiser (models/denoiser/model.py:32-134): no priors, HCAs or IDC
    bool cr = false;                          // true: this context holds the CoarseRestoration network and nothing else
    struct CrStage { std::string name; int C, H, nblk, samp, level; };   // samp: 0 none, 1 down, 2 up
    std::vector<CrStage> cr_stages;
    std::vector<BlockW> cr_blocks;            // execution order
    PackedW cr_samp[9];                       // per stage: down (2x2 s2) or up (1x1 + PixelShuffle) conv
    float* cr_ln_pack = nullptr;
    float* cr_skip[5] = {};                   // encoder-stage outputs kept for the decoder adds (levels 1..4)
    float *cr_loc1 = nullptr, *cr_loc2 = nullptr, *cr_theta = nullptr;   // STN temporaries
    std::vector<Op> cr_program;
    const float* cr_in = nullptr; float* cr_out = nullptr;
    std::string err;
    std::unordered_map<std::string, RawTensor> raw;
    std::vector<void*> allocs;
    std::vector<void*> ws_allocs;             // allocations of the active batch workspace (dev_alloc while ws_scope)
    bool ws_scope = false;
    std::map<int, SavedWs> ws_cache;          // parked workspaces by batch size (at most kWsCached)
    uint64_t ws_clock = 0;
    bool finalized = false;

    // weights
    std::vector<BlockW> den_blocks, fpg_blocks;       // execution order
    std::map<std::string, int> den_block_index;
    PackedW den_down[4], den_up[4], fpg_down[4], fpg_convs[5], idc_conv;
    float *intro_wT = nullptr, *fpg_intro_wT = nullptr, *ending_wT = nullptr;   // intro/ending weights re-laid for coalesced per-lane loads
    HcaW hca[5];
    ResConv res_conv1;
    std::vector<ResBlock> res_blocks;
    int film_total = 0;
    float *film_W = nullptr, *film_b = nullptr, *ln_pack = nullptr, *fpg_ln_pack = nullptr;
    FilmBlock* film_blocks_dev = nullptr;
    float* freq_dev = nullptr;
    int64_t weight_bytes_per_step = 0;
    double flops_per_face_step = 0.0;

    // batch-dependent workspace
    int B = 0;
    std::vector<Chain> chains;
    Chain* ch = nullptr;                     // chain the builder functions currently work on
    float *lat = nullptr, *eps = nullptr;    // [B,4,L,L] of the whole batch; chains own contiguous face ranges
    bool prepared = false;

    // FiLM / schedule
    float *t_dev = nullptr, *temb_a = nullptr, *temb_b = nullptr, *temb_c = nullptr, *film_table = nullptr;
    int film_rows_cap = 0;
    int film_face_stride = 0, film_step_stride = 0;
    bool film_from_cur = false;               // sampling loop: LayerNorm loaders read Chain::film_cur
    float* coef_dev = nullptr;
    int coef_cap = 0;
    int advance = 0;
    hipEvent_t fork_ev = nullptr;
    // hd_sample never blocks on the caller's stream: the schedule is staged through two pinned buffers owned by the
    // context (the one written two calls ago is reused; its copy-done event is the only thing ever waited for), and the
    // FiLM table of a schedule is kept until a different schedule (or hd_eps) overwrites it.
    struct Stage { float* host = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool pending = false; } stage[2];
    int stage_idx = 0;
    std::vector<float> film_sched;            // timesteps whose rows film_table[0..n) currently holds
    bool film_valid = false;
    hipEvent_t film_ev = nullptr;

    // AutoencoderKL context (hd_vae_create)
    bool vae = false;
    struct VaeRes { std::string name; int cin = 0, cout = 0; PackedW c1, c2, sc; bool has_sc = false; const float *n1w = nullptr, *n1b = nullptr, *n2w = nullptr, *n2b = nullptr; };
    struct VaeAttn { std::string name; const float *gw = nullptr, *gb = nullptr; PackedW q, k, v, o; };
    struct VaeW {
        PackedW enc_in, enc_out, dec_in, dec_out, enc_down[3], dec_up[3];
        VaeRes enc_res[4][2], enc_mid[2], dec_mid[2], dec_res[4][3];
        VaeAttn enc_attn, dec_attn;
        const float *enc_nw = nullptr, *enc_nb = nullptr, *dec_nw = nullptr, *dec_nb = nullptr;
        const float *quant_w = nullptr, *quant_b = nullptr, *pq_w = nullptr, *pq_b = nullptr, *ones = nullptr;
    } vw;
    VaeWs vws;
    std::vector<Op> vae_enc_prog, vae_dec_prog;
    const void *vae_enc_key[4] = {}, *vae_dec_key[2] = {};
    int vae_enc_flags = -1;
    uint64_t vae_seed = 0;

    // XCD-local persistent stages (hd_xcd.hpp): latent 16, batch <= 64, one chain, one FiLM row for all faces
    struct XStage {
        XBlockW* blocks_dev = nullptr; unsigned* sync = nullptr; int nblocks = 0;      // sync: flags | hello | gstate, 256 words each
        // autonomous-wave form (hd_xcd2.hpp): the same blocks with the GEMM weights in the 16x16x32 A-operand packing, fragment-order
        // hand-off buffers (64 faces), sync2: flags [8][128] | hello [8][32] | gstate [8][32]
        XBlockW* blocks2_dev = nullptr; unsigned* sync2 = nullptr;
        uint4 *hX = nullptr, *hG = nullptr, *hY = nullptr; float2 *hsx = nullptr, *hsy = nullptr;
    };
    std::map<int, XStage> xstages;            // by index of the stage's first block in den_blocks
    bool xcd_ok = false;                      // the device and the network allow it (setup_xcd)
    bool xcd_on = true;                       // run-time switch (hd_set_option "xcd"): off = the per-GEMM launches of the same program
    int xcd2_mask = 0;                        // levels whose stages also exist in the autonomous-wave form (hd_xcd2.hpp): bit 0 level 2, bit 1 level 3.
                                              // Default 1: measured faster at level 2 (80 vs 90 us for 4 blocks), slower at level 3 (232 vs 217 us
                                              // for 8); HD_XCD2=0..3 overrides (experiments)
    bool xcd2_on = true;                      // run-time switch (hd_set_option "xcd2") between the two forms of the XCD-local stages
    int xcd_phase_limit = 0, xcd_force_global = 0;
    // face-cluster persistent stages of the shallow levels (hd_face.hpp): sync words [flags | gstate] and the pool exchange buffer
    struct FStage { unsigned* sync = nullptr; float* pool_part = nullptr; };
    std::map<int, FStage> fstages;            // by index of the stage's first block
    bool face_ok = false;                     // decided per context in setup_xcd (HD_NO_FACE / HD_NO_XCD at the time the context is finalized)
    bool face_on = true;                      // run-time switch (hd_set_option "face")
    int face_block_limit = 0;
    int face_l1_rows = 16;                    // pixel rows per workgroup of the level-1 stage: 16 = 256 workgroups (every CU), 32 = 128 (HD_FACE_L1_ROWS, experiments)
    int stage_limit_first = -1;               // introspection: the limits above apply only to the stage whose first block has this index (< 0: to all)
    unsigned* xcd_tmo_host = nullptr;         // pinned, device-mapped: non-zero after a hand-off wait gave up
    unsigned* xcd_tmo_dev = nullptr;
    unsigned* abort_dev = nullptr;            // device word: the same code; stage launches read it at entry, poison_if_abort_kernel at the end of a call
    int stage_test_abort = 0;                 // fault injection (hd_set_option "stage_test_abort"): see XStageP / FStageP::test_abort

    // program
    int op_limit = -1, prep_limit = -1;
    bool graphs_valid = false;
    const float* graph_film = nullptr;
    int graph_B = 0;

    // profiling
    bool profiling = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double last_loop_ms = 0.0;
    int last_steps = 0;
    std::map<std::string, std::pair<void*, std::pair<size_t, int>>> dbg;   // name -> (ptr, (elems, is_bf16))
};

#define HD_FAIL(ctx, code, ...)                                   \
    do {                                                          \
        char _b[512];                                             \
        snprintf(_b, sizeof(_b), __VA_ARGS__);                    \
        (ctx)->err = _b;                                          \
        return (code);                                            \
    } while (0)
#define HIPCHECK(ctx, expr)                                                                         \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) HD_FAIL(ctx, HD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

namespace {

template <class T>
int dev_alloc(hd_ctx* c, T** out, size_t count) {
    void* p = nullptr;
    HIPCHECK(c, hipMalloc(&p, count * sizeof(T) + 256));
    (c->ws_scope ? c->ws_allocs : c->allocs).push_back(p);
    *out = reinterpret_cast<T*>(p);
    return HD_OK;
}
void dev_free(hd_ctx* c, void* p) {
    if (!p) return;
    for (auto& a : c->allocs)
        if (a == p) { a = nullptr; break; }
    for (auto& a : c->ws_allocs)
        if (a == p) { a = nullptr; break; }
    (void)hipFree(p);
}

constexpr size_t kWsCached = 3;
void destroy_saved(SavedWs& w) {
    for (auto& ch : w.chains) {
        if (ch.graph_exec) (void)hipGraphExecDestroy(ch.graph_exec);
        if (ch.graph_multi) (void)hipGraphExecDestroy(ch.graph_multi);
        if (ch.stream) (void)hipStreamDestroy(ch.stream);
        if (ch.done) (void)hipEventDestroy(ch.done);
    }
    for (void* p : w.allocs) if (p) (void)hipFree(p);
    w.chains.clear(); w.allocs.clear();
}
// Park the active workspace under its batch size and make the context batch-less (B = 0).
void park_workspace(hd_ctx* c) {
    if (c->B == 0) return;
    SavedWs w;
    w.B = c->B; w.stamp = ++c->ws_clock;
    w.chains = std::move(c->chains); w.lat = c->lat; w.eps = c->eps;
    w.allocs = std::move(c->ws_allocs);
    w.dbg = c->dbg;
    w.graphs_valid = c->graphs_valid; w.graph_film = c->graph_film; w.graph_B = c->graph_B;
    w.cr_program = std::move(c->cr_program); w.cr_in = c->cr_in; w.cr_out = c->cr_out;
    w.vws = c->vws; c->vws = VaeWs();
    w.vae_enc_prog = std::move(c->vae_enc_prog); w.vae_dec_prog = std::move(c->vae_dec_prog);
    c->vae_enc_prog.clear(); c->vae_dec_prog.clear();
    for (auto& k : c->vae_enc_key) k = nullptr;
    for (auto& k : c->vae_dec_key) k = nullptr;
    for (int i = 0; i < 5; ++i) { w.cr_skip[i] = c->cr_skip[i]; c->cr_skip[i] = nullptr; }
    w.cr_loc1 = c->cr_loc1; w.cr_loc2 = c->cr_loc2; w.cr_theta = c->cr_theta;
    c->cr_loc1 = c->cr_loc2 = c->cr_theta = nullptr; c->cr_in = nullptr; c->cr_out = nullptr;
    c->chains.clear(); c->ws_allocs.clear(); c->cr_program.clear();
    c->lat = c->eps = nullptr; c->ch = nullptr;
    for (auto it = c->dbg.begin(); it != c->dbg.end();) it = (it->first == "film" || it->first == "temb") ? std::next(it) : c->dbg.erase(it);
    c->graphs_valid = false; c->prepared = false;
    const int B = c->B;
    c->B = 0;
    c->ws_cache[B] = std::move(w);
    while (c->ws_cache.size() > kWsCached) {                // evict the least recently used (its launches may still be in flight)
        auto old = c->ws_cache.begin();
        for (auto it = c->ws_cache.begin(); it != c->ws_cache.end(); ++it) if (it->second.stamp < old->second.stamp) old = it;
        (void)hipDeviceSynchronize();
        destroy_saved(old->second);
        c->ws_cache.erase(old);
    }
}
// Make the parked workspace of batch B active again; false if there is none.  The conditioning it holds belongs to an
// older batch, so the context is "not prepared" afterwards.
bool unpark_workspace(hd_ctx* c, int B) {
    auto it = c->ws_cache.find(B);
    if (it == c->ws_cache.end()) return false;
    SavedWs& w = it->second;
    c->chains = std::move(w.chains); c->lat = w.lat; c->eps = w.eps;
    c->ws_allocs = std::move(w.allocs);
    for (auto& kv : w.dbg) if (kv.first != "film" && kv.first != "temb") c->dbg[kv.first] = kv.second;
    c->graphs_valid = w.graphs_valid; c->graph_film = w.graph_film; c->graph_B = w.graph_B;
    c->cr_program = std::move(w.cr_program); c->cr_in = w.cr_in; c->cr_out = w.cr_out;
    c->vws = w.vws;                                       // the VAE launch programs are rebuilt (they capture the caller's pointers)
    for (int i = 0; i < 5; ++i) c->cr_skip[i] = w.cr_skip[i];
    c->cr_loc1 = w.cr_loc1; c->cr_loc2 = w.cr_loc2; c->cr_theta = w.cr_theta;
    c->B = B; c->ch = c->chains.empty() ? nullptr : &c->chains[0];
    c->prepared = false;
    c->ws_cache.erase(it);
    return true;
}

const RawTensor* find_raw(hd_ctx* c, const std::string& n) {
    auto it = c->raw.find(n);
    return it == c->raw.end() ? nullptr : &it->second;
}

// ------------------------------------------------------------------------------ manifest (strict load)
// Same keys/shapes as FacialRefiner(latent_res).state_dict() — mirrors hifidiff_amd/arch.py.
using Shape = std::vector<int64_t>;
void m_conv(std::vector<std::pair<std::string, Shape>>& m, const std::string& n, int co, int ci, int kh, int kw, bool bias = true) {
    m.push_back({n + ".weight", {co, ci, kh, kw}});
    if (bias) m.push_back({n + ".bias", {co}});
}
void m_lin(std::vector<std::pair<std::string, Shape>>& m, const std::string& n, int co, int ci) {
    m.push_back({n + ".weight", {co, ci}});
    m.push_back({n + ".bias", {co}});
}
void m_bn(std::vector<std::pair<std::string, Shape>>& m, const std::string& n, int c) {
    m.push_back({n + ".weight", {c}}); m.push_back({n + ".bias", {c}});
    m.push_back({n + ".running_mean", {c}}); m.push_back({n + ".running_var", {c}});
    m.push_back({n + ".num_batches_tracked", {}});
}
void m_naf(std::vector<std::pair<std::string, Shape>>& m, const std::string& p, int c, bool film) {
    m.push_back({p + ".beta", {1, c, 1, 1}}); m.push_back({p + ".gamma", {1, c, 1, 1}});
    if (film) m_lin(m, p + ".mlp.1", 4 * c, FILM_IN);
    m_conv(m, p + ".conv1", 2 * c, c, 1, 1);
    m_conv(m, p + ".conv2", 2 * c, 1, 3, 3);
    m_conv(m, p + ".conv3", c, c, 1, 1);
    m_conv(m, p + ".sca.1", c, c, 1, 1);
    m_conv(m, p + ".conv4", 2 * c, c, 1, 1);
    m_conv(m, p + ".conv5", c, c, 1, 1);
    for (const char* n : {".norm1", ".norm2"}) { m.push_back({p + n + ".weight", {c}}); m.push_back({p + n + ".bias", {c}}); }
}
std::vector<std::pair<std::string, Shape>> build_manifest(int L, bool conditional = true) {
    std::vector<std::pair<std::string, Shape>> m;
    const int enc[4] = {2, 2, 4, 8}, res_layers[4] = {3, 4, 6, 3}, planes[4] = {64, 128, 256, 512};
    // idc
    if (conditional) { m_conv(m, "idc.conv1", 64, 3, 7, 7, false); m_bn(m, "idc.batch_norm1", 64); }
    int cin = 64;
    for (int li = 0; li < 4 && conditional; ++li)
        for (int b = 0; b < res_layers[li]; ++b) {
            std::string q = "idc.layer" + std::to_string(li + 1) + "." + std::to_string(b);
            m_conv(m, q + ".conv1", planes[li], cin, 1, 1); m_bn(m, q + ".batch_norm1", planes[li]);
            m_conv(m, q + ".conv2", planes[li], planes[li], 3, 3); m_bn(m, q + ".batch_norm2", planes[li]);
            m_conv(m, q + ".conv3", planes[li] * 4, planes[li], 1, 1); m_bn(m, q + ".batch_norm3", planes[li] * 4);
            if (b == 0) { m_conv(m, q + ".i_downsample.0", planes[li] * 4, cin, 1, 1); m_bn(m, q + ".i_downsample.1", planes[li] * 4); }
            cin = planes[li] * 4;
        }
    // denoiser
    const std::string d = "denoiser";
    m_lin(m, d + ".time_mlp.1", 1024, 128); m_lin(m, d + ".time_mlp.3", 512, 512);
    m_conv(m, d + ".intro", 128, 4, 3, 3); m_conv(m, d + ".ending", 4, 128, 3, 3);
    int c = WIDTH;
    for (int i = 0; i < 4; ++i) { for (int j = 0; j < enc[i]; ++j) m_naf(m, d + ".encoders." + std::to_string(i) + "." + std::to_string(j), c, true); c *= 2; }
    c = WIDTH * 16;
    for (int i = 0; i < 4; ++i) { c /= 2; for (int j = 0; j < 2; ++j) m_naf(m, d + ".decoders." + std::to_string(i) + "." + std::to_string(j), c, true); }
    c = WIDTH * 16;
    for (int j = 0; j < 8; ++j) m_naf(m, d + ".middle_blks." + std::to_string(j), c, true);
    for (int i = 0; i < 4; ++i) { m_conv(m, d + ".ups." + std::to_string(i) + ".0", c * 2, c, 1, 1, false); c /= 2; }
    c = WIDTH;
    for (int i = 0; i < 4; ++i) { m_conv(m, d + ".downs." + std::to_string(i), 2 * c, c, 2, 2); c *= 2; }
    if (!conditional) return m;                       // Denoiser: time_mlp, intro/ending, blocks, ups, downs only
    c = WIDTH * 16;
    for (int i = 0; i < 5; ++i) {
        std::string p = d + ".hcas." + std::to_string(i);
        m_lin(m, p + ".channel_mlp.0", c, c); m_lin(m, p + ".channel_mlp.2", c, c);
        m_conv(m, p + ".spatial_mlp.0", c / 2, c, 1, 1); m_bn(m, p + ".spatial_mlp.1", c / 2);
        m_conv(m, p + ".spatial_mlp.3", 1, c / 2, 1, 1); m_bn(m, p + ".spatial_mlp.4", 1);
        m_conv(m, p + ".fused_mlp.0", c, c, 3, 3); m_bn(m, p + ".fused_mlp.1", c);
        c /= 2;
    }
    const int s = L / 16;
    m_conv(m, d + ".idc_conv", 2048 * s * s, 2048, 1, 1);
    // fpg
    const std::string f = "fpg";
    m_conv(m, f + ".intro", 128, 4, 3, 3);
    c = WIDTH;
    for (int i = 0; i < 4; ++i) { for (int j = 0; j < enc[i]; ++j) m_naf(m, f + ".encoders." + std::to_string(i) + "." + std::to_string(j), c, false); c *= 2; }
    c = WIDTH;
    for (int i = 0; i < 4; ++i) { m_conv(m, f + ".downs." + std::to_string(i), 2 * c, c, 2, 2); c *= 2; }
    m_conv(m, f + ".convs.0.0", c, c, 1, 1, false);
    for (int i = 1; i < 5; ++i) { m_conv(m, f + ".convs." + std::to_string(i) + ".0", c * 2, c, 1, 1, false); c /= 2; }
    return m;
}

// CoarseRestoration().state_dict() (models/cr/model.py:33-71): mirrors hifidiff_amd/arch.py cr_manifest.
static void cr_stage_list(std::vector<hd_ctx::CrStage>& st) {
    st.clear();
    const int enc[4] = {2, 2, 4, 8};
    int C = 32, H = 128;
    for (int i = 0; i < 4; ++i) { st.push_back({"encoders." + std::to_string(i), C, H, enc[i], 1, i}); C *= 2; H /= 2; }
    st.push_back({"middle_blocks", C, H, 8, 0, 4});
    for (int i = 0; i < 4; ++i) { st.push_back({"decoders." + std::to_string(i), C, H, 2, 2, 4 - i}); C /= 2; H *= 2; }
}
static void stn_shape(int res, int* k0, int* k1, int* fc) {
    if (res <= 8) { *k0 = 3; *k1 = 1; } else if (res <= 16) { *k0 = 5; *k1 = 3; } else if (res <= 32) { *k0 = 7; *k1 = 5; } else { *k0 = 9; *k1 = 7; }
    const int fr = (res - *k0 - 2 * *k1 + 3) / 4;
    *fc = 10 * fr * fr;
}
std::vector<std::pair<std::string, Shape>> build_cr_manifest() {
    std::vector<std::pair<std::string, Shape>> m;
    m_conv(m, "intro", 32, 3, 3, 3); m_conv(m, "outro", 3, 32, 3, 3);
    std::vector<hd_ctx::CrStage> st;
    cr_stage_list(st);
    for (const auto& g : st) {
        for (int j = 0; j < g.nblk; ++j) m_naf(m, g.name + ".nfbs." + std::to_string(j), g.C, false);
        int k0, k1, fc;
        stn_shape(g.H, &k0, &k1, &fc);
        const int n1 = (int)std::sqrt((double)fc);
        m_conv(m, g.name + ".stn.localization.0", 8, g.C, k0, k0); m_conv(m, g.name + ".stn.localization.3", 10, 8, k1, k1);
        m_lin(m, g.name + ".stn.fc_loc.0", n1, fc); m_lin(m, g.name + ".stn.fc_loc.2", 6, n1);
        if (g.samp == 1) m_conv(m, g.name + ".sampling", 2 * g.C, g.C, 2, 2);
        else if (g.samp == 2) m_conv(m, g.name + ".sampling.0", 2 * g.C, g.C, 1, 1, false);
    }
    return m;
}

// ------------------------------------------------------------------------------------------ packing
int upload_vec(hd_ctx* c, const std::vector<float>& v, const float** out) {
    float* d = nullptr;
    int rc = dev_alloc(c, &d, v.size());
    if (rc) return rc;
    HIPCHECK(c, hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    *out = d;
    return HD_OK;
}

// BatchNorm(eval) as y = x*s + o, from host mirrors
int bn_affine(hd_ctx* c, const std::string& bn, std::vector<float>& s, std::vector<float>& o) {
    const RawTensor *w = find_raw(c, bn + ".weight"), *b = find_raw(c, bn + ".bias"), *m = find_raw(c, bn + ".running_mean"),
                    *v = find_raw(c, bn + ".running_var");
    if (!w || !b || !m || !v) HD_FAIL(c, HD_ERR_WEIGHTS, "missing BatchNorm tensors for %s", bn.c_str());
    const size_t n = w->numel;
    s.resize(n); o.resize(n);
    for (size_t i = 0; i < n; ++i) {
        s[i] = w->host[i] / std::sqrt(v->host[i] + 1e-5f);
        o[i] = b->host[i] - m->host[i] * s[i];
    }
    return HD_OK;
}

struct PackOpts { int cin_pad = 0; bool centre_only = false; int S2 = 1; std::string bn; };

// conv/linear weight `name`.weight (+ .bias) -> PackedW (bias folded with BN when opts.bn is set)
int pack_weight(hd_ctx* c, const std::string& name, PackedW* out, const PackOpts& o = PackOpts()) {
    const RawTensor* w = find_raw(c, name + ".weight");
    if (!w) HD_FAIL(c, HD_ERR_WEIGHTS, "missing %s.weight", name.c_str());
    const int N = (int)w->shape[0], Cin = (int)w->shape[1];
    const int KH = w->shape.size() == 4 ? (int)w->shape[2] : 1, KW = w->shape.size() == 4 ? (int)w->shape[3] : 1;
    const int cin_pad = o.cin_pad ? o.cin_pad : Cin;
    const int ntaps = o.centre_only ? 1 : KH * KW;
    PackP p{};
    p.src = w->dev; p.N = N; p.Cin = Cin; p.Cin_pad = cin_pad; p.KH = KH; p.KW = KW; p.ntaps = ntaps;
    p.centre_only = o.centre_only ? 1 : 0; p.S2 = o.S2;
    const int K = ntaps * cin_pad;
    p.Kp = (K + 63) / 64 * 64;
    p.nt_total = (N + 31) / 32;
    std::vector<float> s, off;
    const float* nscale = nullptr;
    if (!o.bn.empty()) {
        int rc = bn_affine(c, o.bn, s, off);
        if (rc) return rc;
        rc = upload_vec(c, s, &nscale);
        if (rc) return rc;
    }
    p.nscale = nscale;
    const size_t n16 = (size_t)p.nt_total * (p.Kp / 16) * 64;
    int rc = dev_alloc(c, &p.dst, n16);
    if (rc) return rc;
    const int blocks = (int)std::min<size_t>((n16 + 255) / 256, 65535);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, 0, p);
    HIPCHECK(c, hipGetLastError());
    out->w = p.dst; out->N = N; out->K = K; out->Kp = p.Kp; out->nt_total = p.nt_total;
    // bias (optionally folded with BN and/or permuted like the output channels)
    const RawTensor* b = find_raw(c, name + ".bias");
    out->bias = b ? b->dev : nullptr;
    if (!o.bn.empty() || (o.S2 > 1 && b)) {
        std::vector<float> bf(N, 0.f);
        for (int n = 0; n < N; ++n) {
            float v = b ? b->host[n] : 0.f;
            if (!o.bn.empty()) v = v * s[n] + off[n];
            bf[n] = v;
        }
        if (o.S2 > 1) {
            std::vector<float> bp(N);
            const int cg = N / o.S2;
            for (int np = 0; np < N; ++np) bp[np] = bf[(np % cg) * o.S2 + np / cg];
            bf.swap(bp);
        }
        rc = upload_vec(c, bf, &out->bias);
        if (rc) return rc;
    }
    return HD_OK;
}

// ------------------------------------------------------------------------------------ GEMM dispatch
// (the launch table lives in hd_dispatch.hpp / hd_dispatch_*.hip, one translation unit per loader family)
bool dwgate_ok(int hw) { return hw == 1 || hw == 4 || hw == 16 || hw == 64 || hw == 256; }

hipError_t dispatch_gemm(const GemmP& p, LdKind lk, EpKind ek, int mode, hipStream_t s) {
    if (lk == LK_LN) return p.film_face_stride != 0 ? dispatch_gemm_ln_face(p, ek, mode, s) : dispatch_gemm_ln_shared(p, ek, mode, s);
    if (lk == LK_BF16 || lk == LK_BF16S) return dispatch_gemm_bf16(p, lk, ek, mode, s);
    return dispatch_gemm_misc(p, lk, ek, mode, s);
}

// Kernel choice: tall tiles while they still give >= 256 workgroups (256 CUs); otherwise the skinny
// kernel (one 32-column weight tile per workgroup, K split across its waves), with 32-row groups when
// 64-row groups would leave most of the chip idle.
int choose_mode(const GemmP& p, bool pair) {
    const int ncols = pair ? p.N / 2 : p.N;
    const int nb64 = (ncols + 63) / 64, nb32 = (ncols + 31) / 32;
    static const int force = hd_env("HD_GEMM_MODE") ? atoi(hd_env("HD_GEMM_MODE")) : -1;
    if (force >= 0) return force;
    const int nb256 = (ncols + (pair ? 127 : 255)) / (pair ? 128 : 256);
    if (p.Kp <= 256 && nb256 <= 2 && ((p.M + 31) / 32) * nb256 >= 512) return 4;   // levels 0/1 at full batch
    // many-row GEMMs: 128/256-row workgroups (waves stacked along M, each with its own K chunks in flight) once K is
    // long (>= 1024), or from K = 512 when 64-row tiles would put four or more workgroups on a CU (latent 32)
    const bool wm_ok = p.Kp >= 1024 || (p.Kp >= 512 && ((p.M + 63) / 64) * nb32 >= 1024);
    if (wm_ok && p.M >= 2048) {
        if (((p.M + 255) / 256) * nb32 >= 256) return 6;
        if (((p.M + 127) / 128) * nb32 >= 192) return 5;
    }
    static const int tall_maxk = hd_env("HD_TALL_MAXK") ? atoi(hd_env("HD_TALL_MAXK")) : 128;   // the tall kernel prefetches one chunk ahead only: with more than two K chunks the skinny kernel (whole K slice in flight) wins even at large M (measured)
    if (p.Kp <= tall_maxk) {
        if (((p.M + 127) / 128) * nb64 >= 256) return 0;
        if (((p.M + 63) / 64) * nb64 >= 256) return 1;
    }
    // 64-row tiles halve the weight re-reads; with a short K (<= 512) and a grid that would only just fill the chip,
    // 32-row tiles (twice the workgroups, two per CU) hide more latency (level 2: conv3/conv5 7.7 -> 6.3 us)
    const int wg64 = ((p.M + 63) / 64) * nb32;
    // (up-convs 2 / 3 and down-conv 0, K <= 512 at 512-1024 such workgroups: 32-row tiles 11.4 / 8.6 / 8.4 us against 14.4 / 10.2 / 9.4)
    static const int short_k_wg = hd_env("HD_SHORTK_WG") ? atoi(hd_env("HD_SHORTK_WG")) : 2048;
    if (wg64 >= 192 && (p.Kp > 512 || wg64 >= short_k_wg)) return 2;
    return 3;
}

GemmP base_gemm(const PackedW& w, int M) {
    GemmP p{};
    p.M = M; p.N = w.N; p.K = w.K; p.Kp = w.Kp; p.nt_total = w.nt_total; p.W = w.w; p.bias = w.bias;
    p.a_scale = 1.f; p.hw = 1; p.ln_eps = 1e-6f; p.shuffle_r = 1; p.stats_np = 1; p.stats_cnt = 1;
    return p;
}

void add_gemm(hd_ctx* c, std::vector<Op>& prog, const std::string& name, GemmP p, LdKind lk, EpKind ek) {
    int t128 = choose_mode(p, ek == EK_GATE || ek == EK_DWGATE);   // (kernel mode; name kept for the capture list)
    {   // many rows, long K, LayerNorm -> gate or bf16 -> residual (latent 32, levels 2 / 3): the deep-prefetch tall kernel; the
        // launch falls back to the mode above when the run-time shape does not fit (per-face timesteps)
        static const bool no_deep = hd_env("HD_NO_DEEP") != nullptr;
        const bool kinds = (lk == LK_LN && ek == EK_GATE) || (lk == LK_BF16 && ek == EK_RESID);
        if (kinds && !no_deep && (ek == EK_GATE ? deep_shape_ok<true>(p) : deep_shape_ok<false>(p))) t128 |= 16;
    }
    {   // each XCD re-fetches what its workgroups read: share the bigger operand through the XCD's L2
        const size_t a_bytes = (size_t)p.M * p.Kp * ((lk == LK_BF16 || lk == LK_BF16S || lk == LK_CONV_BF16 || lk == LK_LN) ? 2 : 4);
        p.xcd_tile_affine = ((size_t)p.N * p.Kp * 2 > a_bytes) ? 1 : 0;
        static const bool no_nt = hd_env("HD_NO_NT") != nullptr;
        static const int nt_maxm = hd_env("HD_NT_MAXM") ? atoi(hd_env("HD_NT_MAXM")) : 256;
        p.w_nt = (!no_nt && p.xcd_tile_affine && p.M <= nt_maxm) ? 1 : 0;     // <= 8 row groups share a weight tile
    }
    const bool film = (lk == LK_LN);
    size_t out_rows = (size_t)p.M * (ek == EK_PIXSHUF ? p.shuffle_r * p.shuffle_r : 1);
    Op op;
    op.name = name; op.out = p.out; op.out_elems = out_rows * p.ldo; op.out_bf16 = (ek == EK_GATE || ek == EK_BIASBF16 || ek == EK_DWGATE) ? 1 : 0;
    Chain* chp = c->ch;
    auto gp = std::make_shared<GemmP>(p);
    op.gemm = gp;
    op.skinny_affine = p.xcd_tile_affine && (t128 & 16) == 0 && t128 != 0 && t128 != 1 && t128 != 4;     // modes 0/1/4 are the tall kernel
    op.run = [c, chp, gp, lk, ek, t128, film](hipStream_t s) mutable -> hipError_t {
                        if (film && gp->film == nullptr) {        // denoiser FiLM rows live in the (re-allocatable) table
                            GemmP q = *gp;
                            q.film = c->film_from_cur ? chp->film_cur : c->film_table;
                            q.film_face_stride = c->film_face_stride;
                            q.film_step_stride = 0;
                            q.step_ptr = nullptr;
                            return dispatch_gemm(q, lk, ek, t128, s);
                        }
                        return dispatch_gemm(*gp, lk, ek, t128, s);
                    };
    prog.push_back(op);
}

// One (Conditional)NAFBlock on level buffers (conditional_naf.py:108-136 / naf.py:105-126): two launches
// where the row-local chain kernel applies (C = 128 / 256), five with the fused conv1 epilogue, seven in the unfused form.
// static_film: FPG blocks use the LayerNorm affine itself as the "FiLM" row (scale = shift = 0).
// x_np/x_cnt: how the LayerNorm partials of the block input X were produced (C/32 x 32 by a GEMM
// epilogue, 1 x C by the intro conv or a skip-add); on return they describe conv5's output.
struct GateOut { const float* gate_c = nullptr; const float* gate_s = nullptr; const float* add = nullptr; };
void add_naf_block(hd_ctx* c, std::vector<Op>& prog, const BlockW& bw, const Level& lv, const float* static_film, int* x_np,
                   int* x_cnt, const GateOut* gate = nullptr) {
    const int C = bw.C, M = lv.M, HW = lv.H * lv.H;
    auto film_fields = [&](GemmP& p, int half) {
        p.hw = HW; p.face0 = c->ch->face0;
        p.film = static_film;                           // nullptr -> patched from the table at launch
        p.film_bias_off = bw.film_off + (2 * half) * C;
        p.film_gain_off = bw.film_off + (2 * half + 1) * C;
        p.film_face_stride = 0; p.film_step_stride = 0; p.step_ptr = nullptr;
    };
    static const bool no_fuse = hd_env("HD_NO_DWFUSE") != nullptr;
    if (dwgate_ok(HW) && !no_fuse) {
        // LN1 + FiLM -> conv1 (+bias) -> depthwise 3x3 -> SimpleGate -> G, pooled mean: one launch
        GemmP p = base_gemm(bw.conv1, M);
        p.A = lv.Xb; p.lda = C; film_fields(p, 0);
        p.stats_in = lv.sx; p.stats_np = *x_np; p.stats_cnt = *x_cnt;
        p.out = lv.G; p.ldo = C; p.dw_w = bw.dw_wT; p.dw_b = bw.dw_b; p.pooled = lv.pooled; p.pooled16 = lv.pooled16; p.side = lv.H;
        add_gemm(c, prog, bw.name + ".conv2_gate_pool", p, LK_LN, EK_DWGATE);
    } else {
        {   // LN1 + FiLM -> conv1 (+bias) -> T1
            GemmP p = base_gemm(bw.conv1, M);
            p.A = lv.Xb; p.lda = C; film_fields(p, 0);
            p.stats_in = lv.sx; p.stats_np = *x_np; p.stats_cnt = *x_cnt;
            p.out = lv.T1; p.ldo = 2 * C;
            add_gemm(c, prog, bw.name + ".conv1", p, LK_LN, EK_BIASF32);
        }
        {   // depthwise 3x3 -> SimpleGate -> G, pooled mean
            const float *T1 = lv.T1, *w = bw.dw_w, *b = bw.dw_b;
            unsigned short* G = lv.G; float* pooled = lv.pooled;
            const int H = lv.H, faces = M / HW;
            float* part = lv.T1 + (size_t)2 * M * C;        // band sums live behind T1 (alloc_chain reserves the room)
            const int nbands = (H + 7) / 8;
            prog.push_back({bw.name + ".conv2_gate_pool", [=](hipStream_t s) -> hipError_t {
                                hipLaunchKernelGGL(dwconv_gate_pool_kernel, dim3(C / 32, faces, nbands), dim3(256), 0, s, T1, w, b, G, part, H, H, C);
                                return hipGetLastError();
                            }});
            prog.back().out = G; prog.back().out_elems = (size_t)M * C; prog.back().out_bf16 = 1;
            prog.push_back({bw.name + ".pool_finish", [=](hipStream_t s) -> hipError_t {
                                hipLaunchKernelGGL(dwconv_pool_finish_kernel, dim3((faces * C + 255) / 256), dim3(256), 0, s, part, pooled, faces, nbands, C,
                                                   1.0f / (float)(H * H));
                                return hipGetLastError();
                            }});
            prog.back().out = pooled; prog.back().out_elems = (size_t)faces * C;
        }
    }
    static const bool no_chain = hd_env("HD_NO_CHAIN") != nullptr;
    if ((C == 128 || C == 256) && HW % 32 == 0 && !no_fuse && !no_chain) {        // also behind the unfused depthwise path (latent 32, level 0)
        // levels 0/1: sca -> conv3 -> residual -> LN+FiLM -> conv4 -> gate -> conv5 -> residual in ONE launch (hd_chain.hpp)
        ChainP q{};
        q.M = M; q.hw = HW; q.face0 = c->ch->face0;
        q.G = lv.G; q.pooled = lv.pooled; q.X = lv.X;
        q.Wsca = bw.sca.w; q.W3 = bw.conv3.w; q.W4 = bw.conv4.w; q.W5 = bw.conv5.w;
        q.bsca = bw.sca.bias; q.b3 = bw.conv3.bias; q.b4 = bw.conv4.bias; q.b5 = bw.conv5.bias; q.beta = bw.beta; q.gamma = bw.gamma;
        q.film = static_film; q.film_bias_off = bw.film_off + 2 * C; q.film_gain_off = bw.film_off + 3 * C; q.ln_eps = 1e-6f;
        q.Xout = lv.X; q.Xout16 = lv.Xb; q.stats_out = lv.sx;
        if (gate) { q.outg16 = lv.Xg; q.gate_c = gate->gate_c; q.gate_s = gate->gate_s; q.add_src = gate->add; q.stats_out = nullptr; q.Xout16 = nullptr; }
        Chain* chp = c->ch;
        const bool big = (C == 256);
        // 64-row workgroups at level 0 halve the weight re-reads but leave one 4-wave workgroup per CU with nothing to
        // overlap its barrier-separated phases with: measured 21.4 us against 16.8 us for 32-row tiles (two per CU) -> opt-in
        static const int mt128 = hd_env("HD_CHAIN_MT") ? atoi(hd_env("HD_CHAIN_MT")) : 1;
        const bool two = !big && mt128 == 2 && HW % 64 == 0 && M % 64 == 0;
        Op op;
        op.name = bw.name + ".conv5"; op.out = lv.X; op.out_elems = (size_t)M * C; op.out_bf16 = 0;
        op.run = [c, chp, q, big, two](hipStream_t s) mutable -> hipError_t {
            ChainP r = q;
            if (r.film == nullptr) {                      // denoiser: FiLM rows live in the (re-allocatable) table
                r.film = c->film_from_cur ? chp->film_cur : c->film_table; r.film_face_stride = c->film_face_stride; r.film_step_stride = 0;
                r.step_ptr = nullptr;
            }
            return big ? launch_chain<256, 1>(r, s) : two ? launch_chain<128, 2>(r, s) : launch_chain<128, 1>(r, s);
        };
        prog.push_back(op);
        *x_np = C / 32; *x_cnt = 32;
        return;
    }
    const bool prescale = dwgate_ok(HW) && !no_fuse && HW <= 16;      // fused conv1 wrote pooled16; few pixels per face (more rows serialise the epilogue)
    if (prescale) {
        {   // SCA on the bf16 pooled vector; its epilogue also scales G in place: G <- bf16(G * s)
            GemmP p = base_gemm(bw.sca, M / HW);
            p.A = lv.pooled16; p.lda = C; p.out = lv.S; p.ldo = C; p.scale_G = lv.G; p.scale_hw = HW;
            add_gemm(c, prog, bw.name + ".sca", p, LK_BF16, EK_SCA);
        }
        {   // conv3 on the pre-scaled G -> y = x + beta * (.)
            GemmP p = base_gemm(bw.conv3, M);
            p.A = lv.G; p.lda = C;
            p.out = lv.Y; p.ldo = C; p.resid = lv.X; p.ldr = C; p.rscale = bw.beta;
            p.stats_out = lv.sy; p.out16 = lv.Yb;
            add_gemm(c, prog, bw.name + ".conv3", p, LK_BF16, EK_RESID);
        }
    } else {
        {   // SCA 1x1 conv on the pooled vector
            GemmP p = base_gemm(bw.sca, M / HW);
            p.out = lv.S; p.ldo = C;
            if (dwgate_ok(HW) && !no_fuse) {            // fused conv1 left a bf16 pooled vector: half the A bytes
                p.A = lv.pooled16; p.lda = C; p.scale_G = lv.G; p.scale_hw = 0;
                add_gemm(c, prog, bw.name + ".sca", p, LK_BF16, EK_SCA);
            } else {
                p.A = lv.pooled; p.lda = C;
                add_gemm(c, prog, bw.name + ".sca", p, LK_F32, EK_BIASF32);
            }
        }
        {   // (G * S) -> conv3 -> y = x + beta * (.)
            GemmP p = base_gemm(bw.conv3, M);
            p.A = lv.G; p.lda = C; p.hw = HW; p.rowscale = lv.S;
            p.out = lv.Y; p.ldo = C; p.resid = lv.X; p.ldr = C; p.rscale = bw.beta;
            p.stats_out = lv.sy; p.out16 = lv.Yb;
            add_gemm(c, prog, bw.name + ".conv3", p, LK_BF16S, EK_RESID);
        }
    }
    {   // LN2 + FiLM -> conv4 -> SimpleGate -> G2 (bf16, reuses G)
        GemmP p = base_gemm(bw.conv4, M);
        p.A = lv.Yb; p.lda = C; film_fields(p, 1);
        p.stats_in = lv.sy; p.stats_np = C / 32; p.stats_cnt = 32;
        p.out = lv.G; p.ldo = C;
        add_gemm(c, prog, bw.name + ".conv4", p, LK_LN, EK_GATE);
    }
    {   // conv5 -> x' = y + gamma * (.)
        GemmP p = base_gemm(bw.conv5, M);
        p.A = lv.G; p.lda = C;
        p.out = lv.X; p.ldo = C; p.resid = lv.Y; p.ldr = C; p.rscale = bw.gamma;
        p.stats_out = lv.sx; p.out16 = lv.Xb;
        if (gate) {                                         // last block before an HCA: also emit the gated conv input
            p.outg16 = lv.Xg; p.gate_c = gate->gate_c; p.gate_s = gate->gate_s; p.add_src = gate->add; p.hw = HW;
            p.stats_out = nullptr; p.out16 = nullptr;       // nothing normalises this tensor next
        }
        add_gemm(c, prog, bw.name + ".conv5", p, LK_BF16, EK_RESID);
    }
    *x_np = C / 32; *x_cnt = 32;
}

void add_down(hd_ctx* c, std::vector<Op>& prog, const std::string& name, const PackedW& w, const Level& src, const Level& dst) {
    GemmP p = base_gemm(w, dst.M);                      // Conv2d(C, 2C, 2, 2) as a patch-gather GEMM
    p.A = src.Xb; p.lda = src.C; p.Hin = src.H; p.Win = src.H; p.Cin = src.C; p.KH = 2; p.KW = 2; p.stride = 2; p.pad = 0;
    p.Hout = dst.H; p.Wout = dst.H; p.ntaps = 4;          // gathers the bf16 copy conv5 wrote (same rounding point as before)
    p.out = dst.X; p.ldo = dst.C; p.stats_out = dst.sx; p.out16 = dst.Xb;
    add_gemm(c, prog, name, p, LK_CONV_BF16, EK_BIASF32);
}

// 1x1 conv (no bias) + PixelShuffle(r) + skip add, written in place over the skip buffer
void add_up(hd_ctx* c, std::vector<Op>& prog, const std::string& name, const PackedW& w, const void* in, bool in_bf16, int M_in,
            int H_in, int C_in, float* out, const float* skip, int r, unsigned short* out16 = nullptr, float2* stats = nullptr) {
    GemmP p = base_gemm(w, M_in);
    p.A = in; p.lda = C_in; p.Hin = H_in; p.Win = H_in; p.shuffle_r = r;
    p.out = out; p.ldo = w.N / (r * r); p.resid = skip; p.bias = nullptr;
    p.out16 = out16; p.stats_out = stats;
    add_gemm(c, prog, name, p, in_bf16 ? LK_BF16 : LK_F32, EK_PIXSHUF);
}

// HCA conv on the pre-gated bf16 tensor Xg that the preceding conv5 epilogue wrote (hca.py:28-29,21-23): a
// plain bf16 implicit GEMM; BN folded, ReLU; fp32 output (+ bf16 copy for the up-conv that follows).
void add_hca(hd_ctx* c, std::vector<Op>& prog, const std::string& name, const HcaW& hw, const unsigned short* in, float* out,
             unsigned short* out16, int M, int H) {
    static const bool no_lds = hd_env("HD_NO_CONVLDS") != nullptr;
    if (!hw.centre_only && !no_lds) {
        // faces small enough to sit in LDS: the A operand is built from the staged faces (hd_conv.hpp)
        ConvP q{};
        q.M = M; q.X = in; q.W = hw.fused.w; q.bias = hw.fused.bias; q.out = out; q.out16 = out16;
        const int C = hw.C;
        int which = -1;
        if (C == 128 && H == 16) which = 0; else if (C == 256 && H == 8) which = 1; else if (C == 512 && H == 4) which = 2;
        else if (C == 1024 && H == 2) which = 3; else if (C == 256 && H == 16) which = 4; else if (C == 512 && H == 8) which = 5;
        else if (C == 1024 && H == 4) which = 6; else if (C == 2048 && H == 2) which = 7; else if (C == 128 && H == 32) which = 8;
        if (which >= 0) {
            Op op;
            op.name = name; op.out = out; op.out_elems = (size_t)M * C;
            op.run = [q, which](hipStream_t s) -> hipError_t {
                switch (which) {
                    case 0: return launch_hca_conv<ConvL0>(q, s);
                    case 1: return launch_hca_conv<ConvL1>(q, s);
                    case 2: return launch_hca_conv<ConvL2>(q, s);
                    case 3: return launch_hca_conv<ConvL3>(q, s);
                    case 4: return launch_hca_conv<ConvL1x32>(q, s);
                    case 5: return launch_hca_conv<ConvL2x32>(q, s);
                    case 6: return launch_hca_conv<ConvL3x32>(q, s);
                    case 8: return launch_hca_conv<ConvL0x32>(q, s);
                    default: return launch_hca_conv<ConvL4x32>(q, s);
                }
            };
            prog.push_back(op);
            return;
        }
    }
    GemmP p = base_gemm(hw.fused, M);
    p.A = in; p.lda = hw.C; p.Hin = H; p.Win = H; p.Cin = hw.C; p.Hout = H; p.Wout = H; p.stride = 1;
    if (hw.centre_only) { p.KH = 1; p.KW = 1; p.pad = 0; p.ntaps = 1; }
    else { p.KH = 3; p.KW = 3; p.pad = 1; p.ntaps = 9; }
    p.out = out; p.ldo = hw.C; p.act = 1; p.out16 = out16;
    add_gemm(c, prog, name, p, LK_CONV_BF16, EK_BIASF32);
}

// Each skinny GEMM can touch the weight tiles of the next weight-dominant skinny GEMM of the program (wrapping
// around: the program is replayed every diffusion step), see prefetch_issue in hd_gemm.hpp.  Measured (r01):
// consumers get 0.5-1 us faster (L2-latency instead of HBM-latency ingest) but the producers, themselves ingest
// bound with no idle memory phase, slow down by more (1592 -> 1633 us/step), so it is OFF unless HD_PREFETCH is set.
void link_prefetch(std::vector<Op>& prog, bool wrap) {
    static const bool off = hd_env("HD_PREFETCH") == nullptr;
    static const std::string only = hd_env("HD_PREFETCH") ? hd_env("HD_PREFETCH") : "";   // "1": every GEMM; else producer-name suffix
    static const size_t min_bytes = hd_env("HD_PF_MIN") ? (size_t)atol(hd_env("HD_PF_MIN")) : (size_t)1 << 20;
    const int n = (int)prog.size();
    for (int i = 0; i < n; ++i) {
        if (!prog[i].gemm) continue;
        prog[i].gemm->pf_base = nullptr;
        if (off) continue;
        if (only != "1" && (prog[i].name.size() < only.size() || prog[i].name.compare(prog[i].name.size() - only.size(), only.size(), only) != 0)) continue;
        // the very next launch only: data touched earlier would be evicted by the launches in between
        const int j = (i + 1 < n) ? i + 1 : (wrap ? 0 : -1);
        if (j < 0 || !prog[j].gemm || !prog[j].skinny_affine) continue;
        const GemmP& nx = *prog[j].gemm;
        const size_t tile_bytes = (size_t)nx.Kp * 32 * 2;
        if ((size_t)nx.nt_total * tile_bytes < min_bytes) continue;
        prog[i].gemm->pf_base = nx.W; prog[i].gemm->pf_tile_u4 = (unsigned)(tile_bytes / 16); prog[i].gemm->pf_ntiles = nx.nt_total;
    }
}

int run_ops(hd_ctx* c, std::vector<Op>& prog, hipStream_t s, int limit = -1) {
    int n = 0;
    for (auto& op : prog) {
        if (limit >= 0 && n >= limit) break;
        hipError_t e = op.run(s);
        if (e != hipSuccess) HD_FAIL(c, HD_ERR_HIP, "launch of %s failed: %s", op.name.c_str(), hipGetErrorString(e));
        ++n;
    }
    return HD_OK;
}


// =============================================================================== CoarseRestoration (§8 f1)
// models/cr/model.py:73-88: intro -> 4 x [NAF blocks, STN, down] (stage outputs are the skips) -> [8 NAF, STN]
// -> 4 x [(+ skip), NAF blocks, STN, up] -> outro.  Stage s works on level buffers of its own geometry
// (C = 32 << level, side 128 >> level); NAF blocks, down- and up-convs are the refiner path's launches.
// conv2.weight [2C][1][3][3] -> tap-major device copy for the fused conv1 epilogue
static int make_dw_layout(hd_ctx* c, BlockW& bw) {
    float* t = nullptr;
    int rc = dev_alloc(c, &t, (size_t)2 * bw.C * 9);
    if (rc) return rc;
    hipLaunchKernelGGL(dw_weight_layout_kernel, dim3((2 * bw.C * 9 + 255) / 256), dim3(256), 0, 0, bw.dw_w, t, 2 * bw.C);
    HIPCHECK(c, hipGetLastError());
    bw.dw_wT = t;
    return HD_OK;
}

static int load_naf_block(hd_ctx* c, const std::string& p, int C, BlockW& bw) {
    bw.name = p; bw.C = C;
    int r = 0;
    r |= pack_weight(c, p + ".conv1", &bw.conv1); r |= pack_weight(c, p + ".conv3", &bw.conv3);
    r |= pack_weight(c, p + ".sca.1", &bw.sca); r |= pack_weight(c, p + ".conv4", &bw.conv4);
    r |= pack_weight(c, p + ".conv5", &bw.conv5);
    if (r) return r;
    bw.dw_w = find_raw(c, p + ".conv2.weight")->dev; bw.dw_b = find_raw(c, p + ".conv2.bias")->dev;
    bw.beta = find_raw(c, p + ".beta")->dev; bw.gamma = find_raw(c, p + ".gamma")->dev;
    return make_dw_layout(c, bw);
}

static int finalize_cr(hd_ctx* c) {
    const auto man = build_cr_manifest();
    for (const auto& e : man) {
        const RawTensor* r = find_raw(c, e.first);
        if (!r) HD_FAIL(c, HD_ERR_WEIGHTS, "Missing key in state_dict: %s", e.first.c_str());
        if (r->shape != e.second) HD_FAIL(c, HD_ERR_WEIGHTS, "size mismatch for %s", e.first.c_str());
    }
    if (c->raw.size() != man.size()) {
        std::unordered_map<std::string, int> known;
        for (const auto& e : man) known[e.first] = 1;
        for (const auto& kv : c->raw)
            if (!known.count(kv.first)) HD_FAIL(c, HD_ERR_WEIGHTS, "Unexpected key in state_dict: %s", kv.first.c_str());
    }
    cr_stage_list(c->cr_stages);
    int rc = 0, off = 0;
    for (size_t si = 0; si < c->cr_stages.size(); ++si) {
        const auto& g = c->cr_stages[si];
        for (int j = 0; j < g.nblk; ++j) {
            BlockW bw;
            rc = load_naf_block(c, g.name + ".nfbs." + std::to_string(j), g.C, bw);
            if (rc) return rc;
            bw.film_off = off; off += 4 * g.C;
            c->cr_blocks.push_back(bw);
        }
        if (g.samp == 1) rc = pack_weight(c, g.name + ".sampling", &c->cr_samp[si]);
        else if (g.samp == 2) { PackOpts o; o.S2 = 4; rc = pack_weight(c, g.name + ".sampling.0", &c->cr_samp[si], o); }   // sub-pixel major (EpPixShufF32)
        if (rc) return rc;
    }
    rc = dev_alloc(c, &c->cr_ln_pack, (size_t)off);
    if (rc) return rc;
    const char* names[4] = {".norm1.bias", ".norm1.weight", ".norm2.bias", ".norm2.weight"};     // table layout [bias | gain] per norm
    for (const BlockW& bw : c->cr_blocks)
        for (int q = 0; q < 4; ++q)
            HIPCHECK(c, hipMemcpy(c->cr_ln_pack + bw.film_off + q * bw.C, find_raw(c, bw.name + names[q])->dev, bw.C * sizeof(float), hipMemcpyDeviceToDevice));
    HIPCHECK(c, hipDeviceSynchronize());
    c->finalized = true;
    return HD_OK;
}

static void add_stn(hd_ctx* c, std::vector<Op>& prog, const std::string& name, const Level& lv, int B) {
    int k0, k1, fc;
    stn_shape(lv.H, &k0, &k1, &fc);
    const int n1 = (int)std::sqrt((double)fc);
    const int H1p = (lv.H - k0 + 1) / 2, H2p = (H1p - k1 + 1) / 2;
    const float *w0 = find_raw(c, name + ".localization.0.weight")->dev, *b0 = find_raw(c, name + ".localization.0.bias")->dev;
    const float *w3 = find_raw(c, name + ".localization.3.weight")->dev, *b3 = find_raw(c, name + ".localization.3.bias")->dev;
    const float *f0w = find_raw(c, name + ".fc_loc.0.weight")->dev, *f0b = find_raw(c, name + ".fc_loc.0.bias")->dev;
    const float *f2w = find_raw(c, name + ".fc_loc.2.weight")->dev, *f2b = find_raw(c, name + ".fc_loc.2.bias")->dev;
    float *loc1 = c->cr_loc1, *loc2 = c->cr_loc2, *theta = c->cr_theta;
    const float* X = lv.X; float* Y = lv.Y; unsigned short* Yb = lv.Yb;
    const int C = lv.C, H = lv.H;
    {   // localization[0..2]: conv k0 (valid) on the channels-last map -> maxpool 2 -> relu
        StnConvP q{};
        q.in = X; q.sb = (long long)H * H * C; q.sc = 1; q.sy = (long long)H * C; q.sx = C;
        q.w = w0; q.bias = b0; q.out = loc1; q.B = B; q.Cin = C; q.Hin = H; q.k = k0; q.Cout = 8; q.Hp = H1p;
        const long long total = (long long)B * 8 * H1p * H1p;
        prog.push_back({name + ".localization.0", [=](hipStream_t s) -> hipError_t {
                            hipLaunchKernelGGL(stn_conv_pool_relu_kernel<8>, dim3((unsigned)((total / 8 + 3) / 4)), dim3(256), 0, s, q);
                            return hipGetLastError();
                        }});
        prog.back().out = loc1; prog.back().out_elems = (size_t)total;
    }
    {   // localization[3..5]: conv k1 on the NCHW result -> maxpool 2 -> relu
        StnConvP q{};
        q.in = loc1; q.sb = (long long)8 * H1p * H1p; q.sc = (long long)H1p * H1p; q.sy = H1p; q.sx = 1;
        q.w = w3; q.bias = b3; q.out = loc2; q.B = B; q.Cin = 8; q.Hin = H1p; q.k = k1; q.Cout = 10; q.Hp = H2p;
        const long long total = (long long)B * 10 * H2p * H2p;
        prog.push_back({name + ".localization.3", [=](hipStream_t s) -> hipError_t {
                            hipLaunchKernelGGL(stn_conv_pool_relu_kernel<10>, dim3((unsigned)((total / 10 + 3) / 4)), dim3(256), 0, s, q);
                            return hipGetLastError();
                        }});
        prog.back().out = loc2; prog.back().out_elems = (size_t)total;
    }
    prog.push_back({name + ".theta", [=](hipStream_t s) -> hipError_t {
                        hipLaunchKernelGGL(stn_fc_kernel, dim3(B), dim3(256), 0, s, loc2, fc, f0w, f0b, n1, f2w, f2b, theta);
                        return hipGetLastError();
                    }});
    prog.back().out = theta; prog.back().out_elems = (size_t)B * 6;
    prog.push_back({name, [=](hipStream_t s) -> hipError_t {
                        const size_t n = (size_t)B * H * H * (C / 4);
                        hipLaunchKernelGGL(stn_grid_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, X, theta, Y, Yb, B, H, C);
                        return hipGetLastError();
                    }});
    prog.back().out = Y; prog.back().out_elems = (size_t)lv.M * C;
}

static int alloc_cr_new(hd_ctx* c, int B);
static int alloc_cr(hd_ctx* c, int B) {
    if (B == c->B) return HD_OK;
    park_workspace(c);
    if (unpark_workspace(c, B)) return HD_OK;
    c->ws_scope = true;
    const int rc = alloc_cr_new(c, B);
    c->ws_scope = false;
    if (rc) { c->B = B; park_workspace(c); auto it = c->ws_cache.find(B); if (it != c->ws_cache.end()) { destroy_saved(it->second); c->ws_cache.erase(it); } }
    return rc;
}
static int alloc_cr_new(hd_ctx* c, int B) {
    c->chains.resize(1);
    Chain& ch = c->chains[0];
    ch.index = 0; ch.B = B; ch.face0 = 0;
    int rc = 0;
    for (int l = 0; l < 5; ++l) {
        Level& v = ch.lv[l];
        v.C = 32 << l; v.H = 128 >> l; v.M = B * v.H * v.H;
        const size_t mc = (size_t)v.M * v.C;
        rc |= dev_alloc(c, &v.X, mc); rc |= dev_alloc(c, &v.Y, mc); rc |= dev_alloc(c, &v.T1, 2 * mc + (size_t)B * ((v.H + 7) / 8) * v.C);
        rc |= dev_alloc(c, &v.G, mc); rc |= dev_alloc(c, &v.pooled, (size_t)B * v.C); rc |= dev_alloc(c, &v.S, (size_t)B * v.C);
        rc |= dev_alloc(c, &v.sx, (size_t)v.M * (v.C / 32)); rc |= dev_alloc(c, &v.sy, (size_t)v.M * (v.C / 32));
        rc |= dev_alloc(c, &v.Xb, mc); rc |= dev_alloc(c, &v.Yb, mc); rc |= dev_alloc(c, &v.Xg, 64);
        rc |= dev_alloc(c, &v.pooled16, (size_t)B * v.C);
        if (l > 0) rc |= dev_alloc(c, &c->cr_skip[l], mc);
        if (rc) return rc;
    }
    // STN temporaries sized for the largest stage (side 128: 8 x 60 x 60 and 10 x 27 x 27 per face)
    rc |= dev_alloc(c, &c->cr_loc1, (size_t)B * 8 * 60 * 60); rc |= dev_alloc(c, &c->cr_loc2, (size_t)B * 10 * 27 * 27);
    rc |= dev_alloc(c, &c->cr_theta, (size_t)B * 6);
    rc |= dev_alloc(c, &ch.step_state, 1);
    if (rc) return rc;
    HIPCHECK(c, hipMemset(ch.step_state, 0, sizeof(StepState)));
    c->B = B;
    c->ch = &ch;
    return HD_OK;
}

// in / out: [B,3,128,128] fp32 NCHW device pointers of this call (captured by the first and last op)
static int build_cr_program(hd_ctx* c, const float* in, float* out) {
    std::vector<Op>& prog = c->cr_program;
    prog.clear();
    Chain& ch = c->chains[0];
    const int B = ch.B;
    {
        const float *w = find_raw(c, "intro.weight")->dev, *b = find_raw(c, "intro.bias")->dev;
        float* X = ch.lv[0].X; unsigned short* Xb = ch.lv[0].Xb; float2* sx = ch.lv[0].sx;
        const size_t M = (size_t)ch.lv[0].M;
        prog.push_back({"intro", [=](hipStream_t s) -> hipError_t {
                            hipLaunchKernelGGL(cr_intro_kernel, dim3((unsigned)((M + 7) / 8)), dim3(256), 0, s, in, w, b, X, Xb, sx, B, 128);
                            return hipGetLastError();
                        }});
        prog.back().out = X; prog.back().out_elems = M * 32;
    }
    int np = 1, cnt = 32, bi = 0;
    for (size_t si = 0; si < c->cr_stages.size(); ++si) {
        const auto& g = c->cr_stages[si];
        const Level& lv = ch.lv[g.level];
        if (si == 5) {
            // decoders.0 input: middle output (STN result in Y) + skip of level 4 (model.py:82-83); later decoder inputs
            // get their skip added by the preceding up-conv epilogue
            const float *A = lv.Y, *S = c->cr_skip[4]; float* X = lv.X; unsigned short* Xb = lv.Xb; float2* sx = lv.sx;
            const int M = lv.M, C = lv.C;
            prog.push_back({g.name + ".skip_add", [=](hipStream_t s) -> hipError_t {
                                hipLaunchKernelGGL(add_rows_stats_kernel, dim3((M + 3) / 4), dim3(256), 0, s, A, S, X, Xb, sx, M, C);
                                return hipGetLastError();
                            }});
            prog.back().out = X; prog.back().out_elems = (size_t)M * C;
            np = 1; cnt = C;
        }
        for (int j = 0; j < g.nblk; ++j) add_naf_block(c, prog, c->cr_blocks[bi++], lv, c->cr_ln_pack, &np, &cnt);
        add_stn(c, prog, g.name + ".stn", lv, B);
        if (g.samp == 1) {
            const Level& dst = ch.lv[g.level + 1];
            GemmP p = base_gemm(c->cr_samp[si], dst.M);           // Conv2d(C, 2C, 2, 2) on the STN output (bf16 copy in Yb)
            p.A = lv.Yb; p.lda = lv.C; p.Hin = lv.H; p.Win = lv.H; p.Cin = lv.C; p.KH = 2; p.KW = 2; p.stride = 2; p.pad = 0;
            p.Hout = dst.H; p.Wout = dst.H; p.ntaps = 4;
            p.out = dst.X; p.ldo = dst.C; p.stats_out = dst.sx; p.out16 = dst.Xb;
            add_gemm(c, prog, g.name, p, LK_CONV_BF16, EK_BIASF32);
            float* skip = c->cr_skip[g.level + 1]; const float* src = dst.X; const size_t bytes = (size_t)dst.M * dst.C * sizeof(float);
            prog.push_back({g.name + ".skip_copy", [=](hipStream_t s) -> hipError_t { return hipMemcpyAsync(skip, src, bytes, hipMemcpyDeviceToDevice, s); }});
            np = dst.C / 32; cnt = 32;
        } else if (g.samp == 2) {
            const Level& lo = ch.lv[g.level - 1];
            add_up(c, prog, g.name, c->cr_samp[si], lv.Yb, true, lv.M, lv.H, lv.C, lo.X, g.level - 1 >= 1 ? c->cr_skip[g.level - 1] : nullptr, 2, lo.Xb, lo.sx);
            np = lo.C / 32; cnt = 32;
        } else {
            prog.back().name = g.name;                             // middle stage: its output is the STN result
        }
    }
    {
        const float *X = ch.lv[0].X, *w = find_raw(c, "outro.weight")->dev, *b = find_raw(c, "outro.bias")->dev;
        const size_t M = (size_t)ch.lv[0].M;
        prog.push_back({"outro", [=](hipStream_t s) -> hipError_t {
                            hipLaunchKernelGGL(cr_outro_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, X, w, b, out, B, 128);
                            return hipGetLastError();
                        }});
        prog.back().out = out; prog.back().out_elems = M * 3;
    }
    c->cr_in = in; c->cr_out = out;
    return HD_OK;
}

// --------------------------------------------------------------------------------------- workspace
int alloc_chain(hd_ctx* c, Chain& ch) {
    const int L = c->L, B = ch.B;
    const bool dbg = (ch.index == 0);                    // introspection reads chain 0
    int rc = 0;
    for (int l = 0; l < 5; ++l) {
        Level& v = ch.lv[l];
        v.C = WIDTH << l; v.H = L >> l; v.M = B * v.H * v.H;
        const size_t mc = (size_t)v.M * v.C;
        rc |= dev_alloc(c, &v.X, mc); rc |= dev_alloc(c, &v.Y, mc); rc |= dev_alloc(c, &v.T1, 2 * mc + (size_t)B * ((v.H + 7) / 8) * v.C);   // + per-band pool sums of the unfused depthwise path
        rc |= dev_alloc(c, &v.G, mc); rc |= dev_alloc(c, &v.pooled, (size_t)B * v.C); rc |= dev_alloc(c, &v.S, (size_t)B * v.C);
        rc |= dev_alloc(c, &v.sx, (size_t)v.M * (v.C / 32)); rc |= dev_alloc(c, &v.sy, (size_t)v.M * (v.C / 32));
        rc |= dev_alloc(c, &v.Xb, mc); rc |= dev_alloc(c, &v.Yb, mc); rc |= dev_alloc(c, &v.Xg, mc);
        rc |= dev_alloc(c, &v.pooled16, (size_t)B * v.C);
        const int pi = 4 - l;                            // prior index: coarsest first
        rc |= dev_alloc(c, &ch.prior[pi], mc); rc |= dev_alloc(c, &ch.gate_c[pi], (size_t)B * v.C);
        rc |= dev_alloc(c, &ch.gate_s[pi], (size_t)v.M);
        if (rc) return rc;
        if (dbg) {
            const std::string s = std::to_string(l);
            c->dbg["X" + s] = {v.X, {mc, 0}}; c->dbg["Y" + s] = {v.Y, {mc, 0}}; c->dbg["T1_" + s] = {v.T1, {2 * mc, 0}};
            c->dbg["G" + s] = {v.G, {mc, 1}}; c->dbg["pooled" + s] = {v.pooled, {(size_t)B * v.C, 0}}; c->dbg["S" + s] = {v.S, {(size_t)B * v.C, 0}};
            c->dbg["Xb" + s] = {v.Xb, {mc, 1}}; c->dbg["Yb" + s] = {v.Yb, {mc, 1}}; c->dbg["Xg" + s] = {v.Xg, {mc, 1}};
            c->dbg["pooled16_" + s] = {v.pooled16, {(size_t)B * v.C, 1}};
            c->dbg["sx" + s] = {v.sx, {(size_t)v.M * (v.C / 32) * 2, 0}}; c->dbg["sy" + s] = {v.sy, {(size_t)v.M * (v.C / 32) * 2, 0}};
            const std::string ps = std::to_string(pi);
            c->dbg["prior" + ps] = {ch.prior[pi], {mc, 0}}; c->dbg["wc" + ps] = {ch.gate_c[pi], {(size_t)B * v.C, 0}};
            c->dbg["ws" + ps] = {ch.gate_s[pi], {(size_t)v.M, 0}};
        }
    }
    rc |= dev_alloc(c, &ch.idc_term, (size_t)B * 2048 * c->S * c->S); rc |= dev_alloc(c, &ch.id_emb, (size_t)B * 2048);
    rc |= dev_alloc(c, &ch.pool_tmp, (size_t)B * 2048); rc |= dev_alloc(c, &ch.mlp_tmp, (size_t)B * 2048);
    rc |= dev_alloc(c, &ch.sp_tmp, (size_t)ch.lv[0].M * 1024);       // >= max over levels of M_l * C_l / 2
    if (rc) return rc;
    if (dbg) { c->dbg["idc"] = {ch.idc_term, {(size_t)B * 2048 * c->S * c->S, 0}}; c->dbg["id_emb"] = {ch.id_emb, {(size_t)B * 2048, 0}}; }
    // ResNet activations (channels-last bf16); largest is conv1 output B x 64x64 x 64 == layer1 B x 32x32 x 256
    const size_t rmax = (size_t)B * 64 * 64 * 64;
    for (int i = 0; i < 4; ++i) rc |= dev_alloc(c, &ch.res_buf[i], rmax);
    rc |= dev_alloc(c, &ch.face8, (size_t)B * 128 * 128);
    rc |= dev_alloc(c, &ch.step_state, 1);
    rc |= dev_alloc(c, &ch.film_cur, (size_t)c->film_total);
    if (rc) return rc;
    HIPCHECK(c, hipMemset(ch.step_state, 0, sizeof(StepState)));
    HIPCHECK(c, hipStreamCreateWithFlags(&ch.stream, hipStreamNonBlocking));
    HIPCHECK(c, hipEventCreateWithFlags(&ch.done, hipEventDisableTiming));
    return HD_OK;
}

int build_denoiser_program(hd_ctx* c);
int get_xstage(hd_ctx* c, int first_block, int nblocks, hd_ctx::XStage** out);
int setup_xcd(hd_ctx* c);

// Cut the batch into chains (HD_CHAINS, default 1).  Two streams of these kernels do overlap (1.6x in
// tools/gemm_bench), but halving M does not make a kernel cheaper, so splitting the batch is not a win.
int alloc_workspace_new(hd_ctx* c, int B);
int alloc_workspace(hd_ctx* c, int B) {
    if (B == c->B) return HD_OK;
    park_workspace(c);                                     // another batch size: keep its buffers, programs and graphs for later
    if (unpark_workspace(c, B)) return HD_OK;
    c->ws_scope = true;
    const int rc = alloc_workspace_new(c, B);
    c->ws_scope = false;
    if (rc) { c->B = B; park_workspace(c); auto it = c->ws_cache.find(B); if (it != c->ws_cache.end()) { destroy_saved(it->second); c->ws_cache.erase(it); } }
    return rc;
}
int alloc_workspace_new(hd_ctx* c, int B) {
    int n = 1;                                            // measured: per-kernel cost barely depends on M, so more chains only add launches
    if (const char* e = hd_env("HD_CHAINS")) n = atoi(e);         // experiment switch (needs HD_EXPERIMENTS=1): measured slower at 2 and 4
    if (n < 1) n = 1;
    if (n > 8) n = 8;
    while (n > 1 && B % n != 0) --n;
    const size_t per_face = (size_t)4 * c->L * c->L;
    int rc = 0;
    rc |= dev_alloc(c, &c->lat, (size_t)B * per_face); rc |= dev_alloc(c, &c->eps, (size_t)B * per_face);
    if (rc) return rc;
    c->dbg["lat"] = {c->lat, {(size_t)B * per_face, 0}}; c->dbg["eps"] = {c->eps, {(size_t)B * per_face, 0}};
    c->chains.resize(n);
    for (int i = 0; i < n; ++i) {
        Chain& ch = c->chains[i];
        ch.index = i; ch.B = B / n; ch.face0 = i * (B / n);
        ch.lat = c->lat + (size_t)ch.face0 * per_face; ch.eps = c->eps + (size_t)ch.face0 * per_face;
        rc = alloc_chain(c, ch);
        if (rc) return rc;
    }
    c->B = B;
    for (auto& ch : c->chains) {
        c->ch = &ch;
        rc = build_denoiser_program(c);
        if (rc) return rc;
    }
    c->ch = &c->chains[0];
    return HD_OK;
}

// ----------------------------------------------------------------------------- program construction
int build_denoiser_program(hd_ctx* c) {
    std::vector<Op>& prog = c->ch->program;
    prog.clear();
    const int B = c->ch->B, L = c->L;
    Chain* chp = c->ch;
    const RawTensor *iw = find_raw(c, "denoiser.intro.weight"), *ib = find_raw(c, "denoiser.intro.bias");
    const RawTensor *ew = find_raw(c, "denoiser.ending.weight"), *eb = find_raw(c, "denoiser.ending.bias");
    {
        const float *lat = c->ch->lat, *w = c->intro_wT, *b = ib->dev; float* out = c->ch->lv[0].X; float2* sx = c->ch->lv[0].sx;
        unsigned short* xb = c->ch->lv[0].Xb;
        const int M = c->ch->lv[0].M;
        prog.push_back({"intro", [=](hipStream_t s) -> hipError_t {
                            if (M >= kLongRunRows) hipLaunchKernelGGL(intro_conv_kernel<16>, dim3((M / 16 + 3) / 4), dim3(256), 0, s, lat, w, b, out, xb, sx, B, L, chp->step_state, c->advance);
                            else hipLaunchKernelGGL(intro_conv_kernel<kIntroPx>, dim3((M / kIntroPx + 3) / 4), dim3(256), 0, s, lat, w, b, out, xb, sx, B, L, chp->step_state, c->advance);
                            return hipGetLastError();
                        }});
        prog.back().out = out; prog.back().out_elems = (size_t)M * 128;
    }
    const int enc[4] = {2, 2, 4, 8};
    int bi = 0;
    int np = 1, cnt = WIDTH;                            // intro emits one (mean, M2) partial per row
    // A run of blocks of one level: as per-GEMM launches, and -- levels 2 and 3 at latent 16, batch <= 64 -- as ONE
    // XCD-local persistent launch (hd_xcd.hpp) when its conditions hold at run time (a single chain: every workgroup
    // must be resident; one FiLM row for all faces).  Both forms compute the same bits.
    int stage_rc = HD_OK;
    auto add_stage = [&](int nblk, const Level& lv, const GateOut* gate) {
        const int first = bi;
        const bool shape_ok = c->xcd_ok && B <= XS_GROUPS * XS_FACES && np == lv.C / 32 && cnt == 32 &&
                              ((lv.C == 1024 && lv.H == 2) || (lv.C == 512 && lv.H == 4)) && nblk <= XS_MAXBLK;
        auto sub = std::make_shared<std::vector<Op>>();
        for (int j = 0; j < nblk; ++j)
            add_naf_block(c, shape_ok ? *sub : prog, c->den_blocks[bi++], lv, nullptr, &np, &cnt, (gate && j == nblk - 1) ? gate : nullptr);
        if (!shape_ok) return;
        hd_ctx::XStage* xs = nullptr;
        stage_rc = get_xstage(c, first, nblk, &xs);
        if (stage_rc) return;
        XStageP sp{};
        sp.B = B; sp.nblocks = nblk; sp.blocks = xs->blocks_dev;
        sp.X = lv.X; sp.Xb = lv.Xb; sp.sx = lv.sx; sp.G = lv.G; sp.Yb = lv.Yb; sp.sy = lv.sy;
        sp.pooled16 = lv.pooled16; sp.pooled = lv.pooled; sp.S = lv.S; sp.ln_eps = 1e-6f;
        if (gate) { sp.outg16 = lv.Xg; sp.gate_c = gate->gate_c; sp.gate_s = gate->gate_s; sp.add_src = gate->add; }
        sp.flags = xs->sync; sp.hello = xs->sync + 256; sp.gstate = xs->sync + 512; sp.tmo = c->xcd_tmo_dev; sp.abort_dev = c->abort_dev;
        const bool l3 = lv.C == 1024;
        X2StageP sp2{};
        if (xs->blocks2_dev) {
            sp2.B = B; sp2.nblocks = nblk; sp2.blocks = xs->blocks2_dev;
            sp2.X = lv.X; sp2.Xb = lv.Xb; sp2.sx = lv.sx; sp2.hX = xs->hX; sp2.hG = xs->hG; sp2.hY = xs->hY; sp2.hsx = xs->hsx; sp2.hsy = xs->hsy;
            sp2.pooled16 = lv.pooled16; sp2.dG = lv.G; sp2.dYb = lv.Yb; sp2.dpooled = lv.pooled; sp2.dS = lv.S; sp2.ln_eps = 1e-6f;
            if (gate) { sp2.outg16 = lv.Xg; sp2.gate_c = gate->gate_c; sp2.gate_s = gate->gate_s; sp2.add_src = gate->add; }
            sp2.flags = xs->sync2; sp2.hello = xs->sync2 + 1024; sp2.gstate = xs->sync2 + 1280; sp2.tmo = c->xcd_tmo_dev; sp2.abort_dev = c->abort_dev;
        }
        const bool have2 = xs->blocks2_dev != nullptr;
        Op op;
        op.name = c->den_blocks[first + nblk - 1].name + ".conv5"; op.out = lv.X; op.out_elems = (size_t)lv.M * lv.C; op.out_bf16 = 0;
        op.run = [c, chp, sp, sp2, have2, sub, l3, first](hipStream_t s) -> hipError_t {
            if (have2 && c->xcd_ok && c->xcd_on && c->xcd2_on && c->chains.size() == 1 && c->film_face_stride == 0) {
                X2StageP r = sp2;
                r.film = c->film_from_cur ? chp->film_cur : c->film_table;
                r.phase_limit = (c->stage_limit_first < 0 || c->stage_limit_first == first) ? c->xcd_phase_limit : 0;
                r.force_global = c->xcd_force_global; r.test_abort = c->stage_test_abort;
                return run_xcd2_stage(l3 ? 1024 : 512, r, s);
            }
            if (c->xcd_ok && c->xcd_on && c->chains.size() == 1 && c->film_face_stride == 0) {
                XStageP r = sp;
                r.film = c->film_from_cur ? chp->film_cur : c->film_table;
                r.phase_limit = (c->stage_limit_first < 0 || c->stage_limit_first == first) ? c->xcd_phase_limit : 0;
                r.force_global = c->xcd_force_global; r.test_abort = c->stage_test_abort;
                return run_xcd_stage(l3 ? 1024 : 512, r, s);
            }
            for (auto& o : *sub) { const hipError_t e = o.run(s); if (e != hipSuccess) return e; }
            return hipSuccess;
        };
        prog.push_back(op);
    };
    // Levels 0 / 1 (latent 16, batch <= 64): a run of blocks as ONE launch with the rows of a face split over a cluster of
    // workgroups (hd_face.hpp); the per-block launches (fused conv1 + chain kernel) stay as the other form of the same op.
    auto add_face_stage = [&](int nblk, const Level& lv, const GateOut* gate, bool want_xb) {
        const int first = bi;
        const bool shape_ok = c->xcd_ok && c->face_ok && B <= 64 && ((lv.C == 128 && lv.H == 16) || (lv.C == 256 && lv.H == 8)) && nblk <= XS_MAXBLK && !(gate && gate->add);
        auto sub = std::make_shared<std::vector<Op>>();
        for (int j = 0; j < nblk; ++j)
            add_naf_block(c, shape_ok ? *sub : prog, c->den_blocks[bi++], lv, nullptr, &np, &cnt, (gate && j == nblk - 1) ? gate : nullptr);
        if (!shape_ok) return;
        hd_ctx::XStage* xs = nullptr;
        stage_rc = get_xstage(c, first, nblk, &xs);
        if (stage_rc) return;
        hd_ctx::FStage& fs = c->fstages[first];
        if (!fs.sync) {
            const bool ws = c->ws_scope;
            c->ws_scope = false;
            int rc = dev_alloc(c, &fs.sync, (size_t)2 * 64 * 16);
            rc |= dev_alloc(c, &fs.pool_part, (size_t)64 * 8 * 256);
            c->ws_scope = ws;
            if (rc || hipMemset(fs.sync, 0, (size_t)2 * 64 * 16 * sizeof(unsigned)) != hipSuccess) { stage_rc = HD_ERR_HIP; return; }
        }
        FStageP fp{};
        fp.B = B; fp.nblocks = nblk; fp.blocks = xs->blocks_dev;
        fp.X = lv.X; fp.Xb = want_xb ? lv.Xb : nullptr; fp.ln_eps = 1e-6f;
        if (gate) { fp.outg16 = lv.Xg; fp.gate_c = gate->gate_c; fp.gate_s = gate->gate_s; }
        fp.pool_part = fs.pool_part; fp.flags = fs.sync; fp.gstate = fs.sync + 64 * 16; fp.tmo = c->xcd_tmo_dev; fp.abort_dev = c->abort_dev;
        const bool c128 = lv.C == 128;
        Op op;
        op.name = c->den_blocks[first + nblk - 1].name + ".conv5"; op.out = lv.X; op.out_elems = (size_t)lv.M * lv.C; op.out_bf16 = 0;
        op.run = [c, chp, fp, sub, c128, first](hipStream_t s) -> hipError_t {
            if (c->xcd_ok && c->face_on && c->chains.size() == 1 && c->film_face_stride == 0) {
                FStageP r = fp;
                r.film = c->film_from_cur ? chp->film_cur : c->film_table;
                r.block_limit = (c->stage_limit_first < 0 || c->stage_limit_first == first) ? c->face_block_limit : 0;
                r.test_abort = c->stage_test_abort;
                const hipError_t e = run_face_stage(c128 ? 128 : 256, c128 ? 32 : c->face_l1_rows, r, s);
                if (e == hipSuccess) return e;
                (void)hipGetLastError();                      // (the dynamic-LDS grant was refused: nothing was launched) -> the per-block launches
                c->face_on = false;
            }
            for (auto& o : *sub) { const hipError_t e = o.run(s); if (e != hipSuccess) return e; }
            return hipSuccess;
        };
        prog.push_back(op);
        np = lv.C / 32; cnt = 32;
    };
    for (int l = 0; l < 4; ++l) {
        if (l >= 2) add_stage(enc[l], c->ch->lv[l], nullptr);
        else add_face_stage(enc[l], c->ch->lv[l], nullptr, true);
        if (stage_rc) return stage_rc;
        add_down(c, prog, "downs." + std::to_string(l), c->den_down[l], c->ch->lv[l], c->ch->lv[l + 1]);
        np = c->ch->lv[l + 1].C / 32; cnt = 32;
    }
    // x + idc_conv(id) -> HCA0 (model.py:245-247): the add and the gate are applied by the last mid block's conv5 epilogue.
    // The unconditional Denoiser (model.py:117-128) has neither: the up-convs read the blocks' own bf16 output.
    const bool cond = c->conditional;
    for (int j = 0; j < 8; ++j) {
        GateOut g0; g0.gate_c = c->ch->gate_c[0]; g0.gate_s = c->ch->gate_s[0]; g0.add = c->ch->idc_term;
        add_naf_block(c, prog, c->den_blocks[bi++], c->ch->lv[4], nullptr, &np, &cnt, (cond && j == 7) ? &g0 : nullptr);
    }
    if (cond) add_hca(c, prog, "hcas.0", c->hca[0], c->ch->lv[4].Xg, c->ch->lv[4].Y, c->ch->lv[4].Yb, c->ch->lv[4].M, c->ch->lv[4].H);
    for (int i = 0; i < 4; ++i) {
        const int l = 3 - i;
        const Level &hi = c->ch->lv[l + 1], &lo = c->ch->lv[l];
        // x = PixelShuffle(up(x)) + enc_skip (model.py:249-251); the epilogue also leaves the bf16 copy and the
        // LayerNorm partials of the new rows (one per 32 channels)
        add_up(c, prog, "ups." + std::to_string(i), c->den_up[i], cond ? hi.Yb : hi.Xb, true, hi.M, hi.H, hi.C, lo.X, lo.X, 2, lo.Xb, lo.sx);
        np = lo.C / 32; cnt = 32;
        GateOut g; g.gate_c = c->ch->gate_c[i + 1]; g.gate_s = c->ch->gate_s[i + 1];
        if (l >= 2) {
            add_stage(2, lo, cond ? &g : nullptr);
            if (stage_rc) return stage_rc;
        } else {
            add_face_stage(2, lo, cond ? &g : nullptr, !cond);          // unconditional: the up conv / ending read the blocks' own output
            if (stage_rc) return stage_rc;
        }
        if (cond) add_hca(c, prog, "hcas." + std::to_string(i + 1), c->hca[i + 1], lo.Xg, lo.Y, i < 3 ? lo.Yb : nullptr, lo.M, lo.H);
    }
    {
        const float *X = cond ? c->ch->lv[0].Y : c->ch->lv[0].X, *w = c->ending_wT, *b = eb->dev; float* eps = c->ch->eps;
        const int M = c->ch->lv[0].M;
        Chain* chp = c->ch;
        prog.push_back({"ending", [=](hipStream_t s) -> hipError_t {
                            // sampling loop (film_from_cur): the launch also applies the scheduler update to this
                            // chain's latents and stages the next step's FiLM row
                            SchedArgs sa{};
                            const bool long_runs = M >= kLongRunRows;
                            unsigned nb = (unsigned)((M / (long_runs ? 16 : kEndingPx) + 3) / 4);
                            if (c->film_from_cur) {
                                const size_t per_face = (size_t)4 * L * L;
                                sa.lat = chp->lat; sa.coef = c->coef_dev; sa.st = chp->step_state;
                                sa.elem0 = (int)(chp->face0 * per_face); sa.n_total = (int)(c->B * per_face);
                                sa.film_table = c->film_table; sa.film_cur = chp->film_cur; sa.film_total = c->film_total;
                                nb += (unsigned)((c->film_total / 4 + 255) / 256);
                            }
                            if (long_runs) hipLaunchKernelGGL(ending_conv_kernel<16>, dim3(nb), dim3(256), 0, s, X, w, b, eps, B, L, sa);
                            else hipLaunchKernelGGL(ending_conv_kernel<kEndingPx>, dim3(nb), dim3(256), 0, s, X, w, b, eps, B, L, sa);
                            return hipGetLastError();
                        }});
        prog.back().out = eps; prog.back().out_elems = (size_t)B * 4 * L * L;
    }
    link_prefetch(prog, true);
    return HD_OK;
}

// HCA gates from prior maps (hca.py:33-48): w_c -> gate_c[i], w_s -> gate_s[i]
void add_gates(hd_ctx* c, std::vector<Op>& prog, int i) {
    const Level& lv = c->ch->lv[4 - i];
    const HcaW& hw = c->hca[i];
    const int C = hw.C, HW = lv.H * lv.H, B = c->ch->B, M = lv.M;
    const float* prior = c->ch->prior[i];
    float *pool = c->ch->pool_tmp, *mlp = c->ch->mlp_tmp, *sp = c->ch->sp_tmp, *gc = c->ch->gate_c[i], *gs = c->ch->gate_s[i];
    const std::string n = "hcas." + std::to_string(i);
    prog.push_back({n + ".pool", [=](hipStream_t s) -> hipError_t {
                        hipLaunchKernelGGL(pool_avgmax_kernel, dim3((C + 255) / 256, B), dim3(256), 0, s, prior, pool, HW, C);
                        return hipGetLastError();
                    }});
    prog.back().out = pool; prog.back().out_elems = (size_t)B * C;
    { GemmP p = base_gemm(hw.mlp0, B); p.A = pool; p.lda = C; p.out = mlp; p.ldo = C; p.act = 1; add_gemm(c, prog, n + ".channel_mlp.0", p, LK_F32, EK_BIASF32); }
    { GemmP p = base_gemm(hw.mlp2, B); p.A = mlp; p.lda = C; p.out = gc; p.ldo = C; p.act = 2; add_gemm(c, prog, n + ".channel_mlp.2", p, LK_F32, EK_BIASF32); }
    { GemmP p = base_gemm(hw.sp0, M); p.A = prior; p.lda = C; p.out = sp; p.ldo = C / 2; p.act = 1; add_gemm(c, prog, n + ".spatial_mlp.0", p, LK_F32, EK_BIASF32); }
    {
        const float* w3 = hw.sp3_w; const float b3 = hw.sp3_b;
        prog.push_back({n + ".spatial_mlp.3", [=](hipStream_t s) -> hipError_t {
                            hipLaunchKernelGGL(rowdot_sigmoid_kernel, dim3((M + 3) / 4), dim3(256), 0, s, sp, w3, b3, gs, M, C / 2);
                            return hipGetLastError();
                        }});
        prog.back().out = gs; prog.back().out_elems = (size_t)M;
    }
}

void add_idc_term(hd_ctx* c, std::vector<Op>& prog) {
    GemmP p = base_gemm(c->idc_conv, c->ch->B);
    p.A = c->ch->id_emb; p.lda = 2048; p.out = c->ch->idc_term; p.ldo = c->idc_conv.N;
    add_gemm(c, prog, "idc_conv", p, LK_F32, EK_BIASF32);
}

void add_fpg(hd_ctx* c, std::vector<Op>& prog, const float* cr_latent_dev) {
    const int B = c->ch->B, L = c->L;
    const RawTensor *iw = find_raw(c, "fpg.intro.weight"), *ib = find_raw(c, "fpg.intro.bias");
    {
        const float *w = c->fpg_intro_wT, *b = ib->dev; float* out = c->ch->lv[0].X; const int M = c->ch->lv[0].M; float2* sx = c->ch->lv[0].sx;
        unsigned short* xb = c->ch->lv[0].Xb;
        Chain* chp = c->ch;
        prog.push_back({"fpg.intro", [=](hipStream_t s) -> hipError_t {
                            if (M >= kLongRunRows) hipLaunchKernelGGL(intro_conv_kernel<16>, dim3((M / 16 + 3) / 4), dim3(256), 0, s, cr_latent_dev, w, b, out, xb, sx, B, L, chp->step_state, 0);
                            else hipLaunchKernelGGL(intro_conv_kernel<kIntroPx>, dim3((M / kIntroPx + 3) / 4), dim3(256), 0, s, cr_latent_dev, w, b, out, xb, sx, B, L, chp->step_state, 0);
                            return hipGetLastError();
                        }});
        prog.back().out = out; prog.back().out_elems = (size_t)M * 128;
    }
    const int enc[4] = {2, 2, 4, 8};
    int bi = 0;
    int np = 1, cnt = WIDTH;
    for (int l = 0; l < 4; ++l) {
        for (int j = 0; j < enc[l]; ++j) add_naf_block(c, prog, c->fpg_blocks[bi++], c->ch->lv[l], c->fpg_ln_pack, &np, &cnt);
        add_down(c, prog, "fpg.downs." + std::to_string(l), c->fpg_down[l], c->ch->lv[l], c->ch->lv[l + 1]);
        np = c->ch->lv[l + 1].C / 32; cnt = 32;
    }
    // convs[0]: 1x1, PixelShuffle(1) == identity -> prior0; then 4x (1x1, PixelShuffle(2), + enc skip)
    add_up(c, prog, "fpg.convs.0", c->fpg_convs[0], c->ch->lv[4].X, false, c->ch->lv[4].M, c->ch->lv[4].H, c->ch->lv[4].C, c->ch->prior[0], nullptr, 1);
    for (int i = 1; i < 5; ++i) {
        const Level &hi = c->ch->lv[5 - i], &lo = c->ch->lv[4 - i];
        // out = shuffled + skip: write into prior[i] with the encoder output as the additive source
        GemmP p = base_gemm(c->fpg_convs[i], hi.M);
        p.A = c->ch->prior[i - 1]; p.lda = hi.C; p.Hin = hi.H; p.Win = hi.H; p.shuffle_r = 2;
        p.out = c->ch->prior[i]; p.ldo = lo.C; p.resid = lo.X; p.bias = nullptr;
        add_gemm(c, prog, "fpg.convs." + std::to_string(i), p, LK_F32, EK_PIXSHUF);
    }
}

void add_resconv(hd_ctx* c, std::vector<Op>& prog, const std::string& name, const ResConv& rc, const unsigned short* in, int Hin,
                 unsigned short* out, const unsigned short* resid, bool relu) {
    const int Hout = (Hin + 2 * rc.pad - rc.k) / rc.stride + 1;
    GemmP p = base_gemm(rc.w, c->ch->B * Hout * Hout);
    p.A = in; p.out = out; p.ldo = rc.cout; p.resid = resid; p.ldr = rc.cout; p.act = relu ? 1 : 0;
    if (rc.k == 1 && rc.stride == 1) {
        p.lda = rc.cin;
        add_gemm(c, prog, name, p, LK_BF16, EK_BIASBF16);
    } else {
        const int cin = rc.w.K / (rc.k * rc.k);          // padded Cin
        p.lda = cin; p.Hin = Hin; p.Win = Hin; p.Cin = cin; p.KH = rc.k; p.KW = rc.k; p.stride = rc.stride; p.pad = rc.pad;
        p.Hout = Hout; p.Wout = Hout; p.ntaps = rc.k * rc.k;
        add_gemm(c, prog, name, p, LK_CONV_BF16, EK_BIASBF16);
    }
}

// ResNet-50 trunk (idc/model.py:122-135) on cr_face -> id_emb [B][2048]
void add_resnet(hd_ctx* c, std::vector<Op>& prog, const float* cr_face_dev) {
    const int B = c->ch->B;
    {
        uint4* f8 = c->ch->face8; const size_t npix = (size_t)B * 128 * 128;
        prog.push_back({"idc.input", [=](hipStream_t s) -> hipError_t {
                            hipLaunchKernelGGL(nchw3_to_nhwc8_bf16_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, cr_face_dev, f8, 128 * 128, npix);
                            return hipGetLastError();
                        }});
        prog.back().out = f8; prog.back().out_elems = npix * 8; prog.back().out_bf16 = 1;
    }
    unsigned short *b0 = c->ch->res_buf[0], *b1 = c->ch->res_buf[1], *b2 = c->ch->res_buf[2], *b3 = c->ch->res_buf[3];
    add_resconv(c, prog, "idc.conv1", c->res_conv1, reinterpret_cast<const unsigned short*>(c->ch->face8), 128, b0, nullptr, true);
    {
        const size_t total = (size_t)B * 32 * 32 * 64;
        prog.push_back({"idc.max_pool", [=](hipStream_t s) -> hipError_t {
                            hipLaunchKernelGGL(maxpool3x3s2_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, b0, b1, B, 64, 64, 64);
                            return hipGetLastError();
                        }});
        prog.back().out = b1; prog.back().out_elems = total; prog.back().out_bf16 = 1;
    }
    unsigned short *x = b1, *y1 = b0, *y2 = b2, *idn = b3;
    int H = 32, bi = 0;
    const int res_layers[4] = {3, 4, 6, 3};
    for (int li = 0; li < 4; ++li)
        for (int b = 0; b < res_layers[li]; ++b) {
            const ResBlock& rb = c->res_blocks[bi++];
            const std::string q = "idc.layer" + std::to_string(li + 1) + "." + std::to_string(b);
            const int Hout = H / rb.c2.stride;
            add_resconv(c, prog, q + ".conv1", rb.c1, x, H, y1, nullptr, true);
            add_resconv(c, prog, q + ".conv2", rb.c2, y1, H, y2, nullptr, true);
            const unsigned short* identity = x;
            if (rb.has_ds) { add_resconv(c, prog, q + ".i_downsample", rb.ds, x, H, idn, nullptr, false); identity = idn; }
            // conv3 + BN + identity -> ReLU, written to y1 (free again), then rotate
            add_resconv(c, prog, q + ".conv3", rb.c3, y2, Hout, y1, identity, true);
            std::swap(x, y1);
            H = Hout;
        }
    {
        float* emb = c->ch->id_emb; const unsigned short* xin = x; const int HW = H * H;
        prog.push_back({"idc.avgpool", [=](hipStream_t s) -> hipError_t {
                            hipLaunchKernelGGL(avgpool_bf16_kernel, dim3(2048 / 256, B), dim3(256), 0, s, xin, emb, HW, 2048);
                            return hipGetLastError();
                        }});
        prog.back().out = emb; prog.back().out_elems = (size_t)B * 2048;
    }
}

// ------------------------------------------------------------------------------------------ FiLM
int ensure_film_rows(hd_ctx* c, int rows) {
    if (rows <= c->film_rows_cap) return HD_OK;
    dev_free(c, c->t_dev); dev_free(c, c->temb_a); dev_free(c, c->temb_b); dev_free(c, c->temb_c); dev_free(c, c->film_table);
    int rc = 0;
    rc |= dev_alloc(c, &c->t_dev, rows); rc |= dev_alloc(c, &c->temb_a, (size_t)rows * 128);
    rc |= dev_alloc(c, &c->temb_b, (size_t)rows * 1024); rc |= dev_alloc(c, &c->temb_c, (size_t)rows * 512);
    rc |= dev_alloc(c, &c->film_table, (size_t)rows * c->film_total);
    if (rc) return rc;
    c->film_rows_cap = rows;
    c->film_valid = false;
    c->dbg["film"] = {c->film_table, {(size_t)rows * c->film_total, 0}};
    c->dbg["temb"] = {c->temb_c, {(size_t)rows * 512, 0}};
    return HD_OK;
}

// t (device, n values) -> FiLM table rows [n][film_total] (gain/bias with LN affine folded)
int compute_film(hd_ctx* c, const float* t_dev, int n, hipStream_t s) {
    const RawTensor *w1 = find_raw(c, "denoiser.time_mlp.1.weight"), *b1 = find_raw(c, "denoiser.time_mlp.1.bias");
    const RawTensor *w3 = find_raw(c, "denoiser.time_mlp.3.weight"), *b3 = find_raw(c, "denoiser.time_mlp.3.bias");
    hipLaunchKernelGGL(time_embed_kernel, dim3((n * 64 + 255) / 256), dim3(256), 0, s, t_dev, c->freq_dev, c->temb_a, n);
    hipLaunchKernelGGL((linear_f32_kernel<false>), dim3(1024 / 64, (n + 63) / 64), dim3(256), 0, s, c->temb_a, 128, w1->dev, b1->dev, c->temb_b, 1024, n, 1024, 128);
    hipLaunchKernelGGL((linear_f32_kernel<true>), dim3(512 / 64, (n + 63) / 64), dim3(256), 0, s, c->temb_b, 1024, w3->dev, b3->dev, c->temb_c, 512, n, 512, 512);
    hipLaunchKernelGGL((linear_f32_kernel<true>), dim3((c->film_total + 63) / 64, (n + 63) / 64), dim3(256), 0, s, c->temb_c, 512, c->film_W, c->film_b,
                       c->film_table, c->film_total, n, c->film_total, FILM_IN);
    hipLaunchKernelGGL(film_fold_kernel, dim3(2048 / 256, (unsigned)c->den_blocks.size(), n), dim3(256), 0, s, c->film_table, c->ln_pack, c->film_blocks_dev, c->film_total);
    HIPCHECK(c, hipGetLastError());
    return HD_OK;
}



// ============================================================================================ VAE boundary
// AutoencoderKL of "stable-diffusion-2-1-base" (test_refiner.py:176-178): block_out_channels (128, 256, 512, 512),
// layers_per_block 2, norm_num_groups 32, latent_channels 4, one attention head of 512 in each mid block
// (diffusers 0.32.2 AutoencoderKL / Encoder / Decoder / ResnetBlock2D / Attention; third party, absent: parity unpinned).
static void m_norm(std::vector<std::pair<std::string, Shape>>& m, const std::string& n, int c) { m.push_back({n + ".weight", {c}}); m.push_back({n + ".bias", {c}}); }
static void m_vres(std::vector<std::pair<std::string, Shape>>& m, const std::string& p, int cin, int cout) {
    m_norm(m, p + ".norm1", cin); m_conv(m, p + ".conv1", cout, cin, 3, 3); m_norm(m, p + ".norm2", cout); m_conv(m, p + ".conv2", cout, cout, 3, 3);
    if (cin != cout) m_conv(m, p + ".conv_shortcut", cout, cin, 1, 1);
}
static void m_vattn(std::vector<std::pair<std::string, Shape>>& m, const std::string& p, int c) {
    m_norm(m, p + ".group_norm", c); m_lin(m, p + ".to_q", c, c); m_lin(m, p + ".to_k", c, c); m_lin(m, p + ".to_v", c, c); m_lin(m, p + ".to_out.0", c, c);
}
static const int kVaeCh[4] = {128, 256, 512, 512};
std::vector<std::pair<std::string, Shape>> build_vae_manifest() {
    std::vector<std::pair<std::string, Shape>> m;
    m_conv(m, "encoder.conv_in", 128, 3, 3, 3);
    int cin = 128;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 2; ++j) { m_vres(m, "encoder.down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), cin, kVaeCh[i]); cin = kVaeCh[i]; }
        if (i < 3) m_conv(m, "encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv", cin, cin, 3, 3);
    }
    m_vres(m, "encoder.mid_block.resnets.0", 512, 512); m_vattn(m, "encoder.mid_block.attentions.0", 512); m_vres(m, "encoder.mid_block.resnets.1", 512, 512);
    m_norm(m, "encoder.conv_norm_out", 512); m_conv(m, "encoder.conv_out", 8, 512, 3, 3);
    m_conv(m, "quant_conv", 8, 8, 1, 1); m_conv(m, "post_quant_conv", 4, 4, 1, 1);
    m_conv(m, "decoder.conv_in", 512, 4, 3, 3);
    m_vres(m, "decoder.mid_block.resnets.0", 512, 512); m_vattn(m, "decoder.mid_block.attentions.0", 512); m_vres(m, "decoder.mid_block.resnets.1", 512, 512);
    cin = 512;
    for (int i = 0; i < 4; ++i) {
        const int cout = kVaeCh[3 - i];
        for (int j = 0; j < 3; ++j) { m_vres(m, "decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), cin, cout); cin = cout; }
        if (i < 3) m_conv(m, "decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv", cin, cin, 3, 3);
    }
    m_norm(m, "decoder.conv_norm_out", 128); m_conv(m, "decoder.conv_out", 3, 128, 3, 3);
    return m;
}

static int finalize_vae(hd_ctx* c) {
    const auto man = build_vae_manifest();
    for (const auto& e : man) {
        const RawTensor* r = find_raw(c, e.first);
        if (!r) HD_FAIL(c, HD_ERR_WEIGHTS, "Missing key in state_dict: %s", e.first.c_str());
        if (r->shape != e.second) HD_FAIL(c, HD_ERR_WEIGHTS, "size mismatch for %s", e.first.c_str());
    }
    if (c->raw.size() != man.size()) {
        std::unordered_map<std::string, int> known;
        for (const auto& e : man) known[e.first] = 1;
        for (const auto& kv : c->raw)
            if (!known.count(kv.first)) HD_FAIL(c, HD_ERR_WEIGHTS, "Unexpected key in state_dict: %s", kv.first.c_str());
    }
    auto& w = c->vw;
    int rc = 0;
    auto res = [&](const std::string& p, int cin, int cout, hd_ctx::VaeRes& r) {
        r.name = p; r.cin = cin; r.cout = cout;
        rc |= pack_weight(c, p + ".conv1", &r.c1); rc |= pack_weight(c, p + ".conv2", &r.c2);
        r.has_sc = cin != cout;
        if (r.has_sc) rc |= pack_weight(c, p + ".conv_shortcut", &r.sc);
        r.n1w = find_raw(c, p + ".norm1.weight")->dev; r.n1b = find_raw(c, p + ".norm1.bias")->dev;
        r.n2w = find_raw(c, p + ".norm2.weight")->dev; r.n2b = find_raw(c, p + ".norm2.bias")->dev;
    };
    auto attn = [&](const std::string& p, hd_ctx::VaeAttn& a) {
        a.name = p; a.gw = find_raw(c, p + ".group_norm.weight")->dev; a.gb = find_raw(c, p + ".group_norm.bias")->dev;
        rc |= pack_weight(c, p + ".to_q", &a.q); rc |= pack_weight(c, p + ".to_k", &a.k); rc |= pack_weight(c, p + ".to_v", &a.v); rc |= pack_weight(c, p + ".to_out.0", &a.o);
    };
    PackOpts pad8; pad8.cin_pad = 8;
    rc |= pack_weight(c, "encoder.conv_in", &w.enc_in, pad8); rc |= pack_weight(c, "decoder.conv_in", &w.dec_in, pad8);
    rc |= pack_weight(c, "encoder.conv_out", &w.enc_out); rc |= pack_weight(c, "decoder.conv_out", &w.dec_out);
    int cin = 128;
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 2; ++j) { res("encoder.down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), cin, kVaeCh[i], w.enc_res[i][j]); cin = kVaeCh[i]; }
        if (i < 3) rc |= pack_weight(c, "encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv", &w.enc_down[i]);
    }
    res("encoder.mid_block.resnets.0", 512, 512, w.enc_mid[0]); attn("encoder.mid_block.attentions.0", w.enc_attn); res("encoder.mid_block.resnets.1", 512, 512, w.enc_mid[1]);
    res("decoder.mid_block.resnets.0", 512, 512, w.dec_mid[0]); attn("decoder.mid_block.attentions.0", w.dec_attn); res("decoder.mid_block.resnets.1", 512, 512, w.dec_mid[1]);
    cin = 512;
    for (int i = 0; i < 4; ++i) {
        const int cout = kVaeCh[3 - i];
        for (int j = 0; j < 3; ++j) { res("decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), cin, cout, w.dec_res[i][j]); cin = cout; }
        if (i < 3) rc |= pack_weight(c, "decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv", &w.dec_up[i]);
    }
    if (rc) return rc;
    w.enc_nw = find_raw(c, "encoder.conv_norm_out.weight")->dev; w.enc_nb = find_raw(c, "encoder.conv_norm_out.bias")->dev;
    w.dec_nw = find_raw(c, "decoder.conv_norm_out.weight")->dev; w.dec_nb = find_raw(c, "decoder.conv_norm_out.bias")->dev;
    w.quant_w = find_raw(c, "quant_conv.weight")->dev; w.quant_b = find_raw(c, "quant_conv.bias")->dev;
    w.pq_w = find_raw(c, "post_quant_conv.weight")->dev; w.pq_b = find_raw(c, "post_quant_conv.bias")->dev;
    rc = upload_vec(c, std::vector<float>(512, 1.0f), &w.ones);
    if (rc) return rc;
    HIPCHECK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&vae_attention_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, AT_SMEM));
    HIPCHECK(c, hipDeviceSynchronize());
    c->finalized = true;
    return HD_OK;
}

static int alloc_vae_new(hd_ctx* c, int B, int R) {
    auto& v = c->vws;
    const size_t px = (size_t)B * R * R, big = px * 256, lat = (size_t)B * (R / 8) * (R / 8);
    int rc = 0;
    rc |= dev_alloc(c, &v.X, big); rc |= dev_alloc(c, &v.T, big); rc |= dev_alloc(c, &v.S, big);
    rc |= dev_alloc(c, &v.H, big); rc |= dev_alloc(c, &v.H2, big); rc |= dev_alloc(c, &v.Xb, big); rc |= dev_alloc(c, &v.U, big);
    rc |= dev_alloc(c, &v.in8, px); rc |= dev_alloc(c, &v.resz, px * 3); rc |= dev_alloc(c, &v.out3, px * 3);
    rc |= dev_alloc(c, &v.mom, lat * 8); rc |= dev_alloc(c, &v.Q, lat * 512); rc |= dev_alloc(c, &v.K, lat * 512); rc |= dev_alloc(c, &v.V, lat * 512);
    rc |= dev_alloc(c, &v.part, (size_t)B * ((size_t)R * R / 256 + 1) * GN_GROUPS * 2);
    v.B = B; v.R = R;
    return rc;
}
static int alloc_vae(hd_ctx* c, int B, int R) {
    const int key = B + 8192 * (R / 8);
    if (key == c->B) return HD_OK;
    park_workspace(c);
    if (unpark_workspace(c, key)) return HD_OK;
    c->ws_scope = true;
    const int rc = alloc_vae_new(c, B, R);
    c->ws_scope = false;
    c->B = key;
    if (rc) { park_workspace(c); auto it = c->ws_cache.find(key); if (it != c->ws_cache.end()) { destroy_saved(it->second); c->ws_cache.erase(it); } }
    return rc;
}

// ---- launch-program pieces (channels-last fp32 residual stream X [B*H*H][C]) ----
static void vae_groupnorm(hd_ctx* c, std::vector<Op>& prog, const std::string& name, const float* x, const float* gw, const float* gb,
                          unsigned short* y, int B, int H, int C, bool silu) {
    const int HW = H * H, chunk = 256, nch = (HW + chunk - 1) / chunk;
    double* part = c->vws.part;
    prog.push_back({name, [=](hipStream_t s) -> hipError_t {
                        hipLaunchKernelGGL(groupnorm_partial_kernel, dim3(nch, B), dim3(256), 0, s, x, part, HW, C, chunk);
                        hipLaunchKernelGGL(groupnorm_apply_kernel, dim3(nch, B), dim3(256), 0, s, x, part, nch, gw, gb, y, HW, C, chunk, 1e-6f, silu ? 1 : 0);
                        return hipGetLastError();
                    }});
    prog.back().out = y; prog.back().out_elems = (size_t)B * HW * C; prog.back().out_bf16 = 1;
}
// 3x3 conv (pad 1; stride 2: Downsample2D pads right/bottom only = reading zeros past the edge) on a bf16 channels-last map
static void vae_conv3(hd_ctx* c, std::vector<Op>& prog, const std::string& name, const PackedW& w, const unsigned short* in, int B, int Hin,
                      int stride, float* out, const float* resid, unsigned short* out16) {
    const int cin = w.K / 9, Hout = Hin / stride;
    GemmP p = base_gemm(w, B * Hout * Hout);
    p.A = in; p.lda = cin; p.Hin = Hin; p.Win = Hin; p.Cin = cin; p.KH = 3; p.KW = 3; p.stride = stride; p.pad = stride == 1 ? 1 : 0;
    p.Hout = Hout; p.Wout = Hout; p.ntaps = 9;
    p.out = out; p.ldo = w.N; p.out16 = out16;
    if (resid) { p.resid = resid; p.ldr = w.N; p.rscale = c->vw.ones; add_gemm(c, prog, name, p, LK_CONV_BF16, EK_RESID); }
    else add_gemm(c, prog, name, p, LK_CONV_BF16, EK_BIASF32);
}
static void vae_resnet(hd_ctx* c, std::vector<Op>& prog, const hd_ctx::VaeRes& r, int B, int H, unsigned short* out16) {
    auto& v = c->vws;
    vae_groupnorm(c, prog, r.name + ".norm1", v.X, r.n1w, r.n1b, v.H, B, H, r.cin, true);
    vae_conv3(c, prog, r.name + ".conv1", r.c1, v.H, B, H, 1, v.T, nullptr, nullptr);
    vae_groupnorm(c, prog, r.name + ".norm2", v.T, r.n2w, r.n2b, v.H2, B, H, r.cout, true);
    const float* resid = v.X;
    if (r.has_sc) {
        GemmP p = base_gemm(r.sc, B * H * H);
        p.A = v.X; p.lda = r.cin; p.out = v.S; p.ldo = r.cout;
        add_gemm(c, prog, r.name + ".conv_shortcut", p, LK_F32, EK_BIASF32);
        resid = v.S;
    }
    vae_conv3(c, prog, r.name + ".conv2", r.c2, v.H2, B, H, 1, v.X, resid, out16);
}
static void vae_attention(hd_ctx* c, std::vector<Op>& prog, const hd_ctx::VaeAttn& a, int B, int H) {
    auto& v = c->vws;
    const int T = H * H, M = B * T;
    vae_groupnorm(c, prog, a.name + ".group_norm", v.X, a.gw, a.gb, v.H, B, H, 512, false);
    const PackedW* ws[3] = {&a.q, &a.k, &a.v}; float* outs[3] = {v.Q, v.K, v.V}; const char* nm[3] = {".to_q", ".to_k", ".to_v"};
    for (int i = 0; i < 3; ++i) {
        GemmP p = base_gemm(*ws[i], M);
        p.A = v.H; p.lda = 512; p.out = outs[i]; p.ldo = 512;
        add_gemm(c, prog, a.name + nm[i], p, LK_BF16, EK_BIASF32);
    }
    {
        const float *q = v.Q, *k = v.K, *vv = v.V; unsigned short* o = v.H2;
        prog.push_back({a.name + ".softmax_qk_v", [=](hipStream_t s) -> hipError_t {
                            hipLaunchKernelGGL(vae_attention_kernel, dim3((T + AT_Q - 1) / AT_Q, B), dim3(256), AT_SMEM, s, q, k, vv, o, T, 1.0f / sqrtf((float)AT_C));
                            return hipGetLastError();
                        }});
        prog.back().out = o; prog.back().out_elems = (size_t)M * 512; prog.back().out_bf16 = 1;
    }
    GemmP p = base_gemm(a.o, M);
    p.A = v.H2; p.lda = 512; p.out = v.X; p.ldo = 512; p.resid = v.X; p.ldr = 512; p.rscale = c->vw.ones;
    add_gemm(c, prog, a.name + ".to_out.0", p, LK_BF16, EK_RESID);
}

static int build_vae_encode(hd_ctx* c, int B, int in_res, int R, const float* images, int vae_range, const float* noise, uint64_t seed,
                            float* moments_out, float* latents_out) {
    auto& prog = c->vae_enc_prog; prog.clear();
    auto& v = c->vws; auto& w = c->vw;
    const float* src = images;
    if (in_res != R) {                                     // F.interpolate(x, R, mode="bicubic", align_corners=False) (test_refiner.py:80)
        float* dst = v.resz; const int planes = B * 3;
        prog.push_back({"bicubic", [=](hipStream_t s) -> hipError_t {
                            const size_t n = (size_t)planes * R * R;
                            hipLaunchKernelGGL(bicubic_resize_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 65535)), dim3(256), 0, s, images, dst, planes, in_res, in_res, R, R);
                            return hipGetLastError();
                        }});
        prog.back().out = dst; prog.back().out_elems = (size_t)planes * R * R;
        src = dst;
    }
    {
        uint4* in8 = v.in8; const size_t npix = (size_t)B * R * R; const int HW = R * R;
        prog.push_back({"encoder.input", [=](hipStream_t s) -> hipError_t {
                            hipLaunchKernelGGL(nchw_to_nhwc8_bf16_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, src, in8, 3, HW, npix, vae_range);
                            return hipGetLastError();
                        }});
    }
    vae_conv3(c, prog, "encoder.conv_in", w.enc_in, reinterpret_cast<const unsigned short*>(v.in8), B, R, 1, v.X, nullptr, nullptr);
    int H = R;
    for (int i = 0; i < 4; ++i) {
        vae_resnet(c, prog, w.enc_res[i][0], B, H, nullptr);
        vae_resnet(c, prog, w.enc_res[i][1], B, H, i < 3 ? v.Xb : nullptr);
        if (i < 3) { vae_conv3(c, prog, "encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv", w.enc_down[i], v.Xb, B, H, 2, v.X, nullptr, nullptr); H /= 2; }
    }
    vae_resnet(c, prog, w.enc_mid[0], B, H, nullptr);
    vae_attention(c, prog, w.enc_attn, B, H);
    vae_resnet(c, prog, w.enc_mid[1], B, H, nullptr);
    vae_groupnorm(c, prog, "encoder.conv_norm_out", v.X, w.enc_nw, w.enc_nb, v.H, B, H, 512, true);
    vae_conv3(c, prog, "encoder.conv_out", w.enc_out, v.H, B, H, 1, v.mom, nullptr, nullptr);
    {
        const float *mom = v.mom, *qw = w.quant_w, *qb = w.quant_b; const int HW = H * H; const size_t npix = (size_t)B * HW;
        float* tmp = v.Q;                                  // quant_conv output when only the moments are wanted
        prog.push_back({"quant_conv.sample", [=](hipStream_t s) -> hipError_t {
                            if (latents_out)
                                hipLaunchKernelGGL(vae_sample_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, mom, qw, qb, noise, (unsigned long long)seed, latents_out, HW, npix, 0.18215f, 8);
                            if (moments_out)
                                hipLaunchKernelGGL(vae_moments_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, mom, qw, qb, moments_out, HW, npix);
                            (void)tmp;
                            return hipGetLastError();
                        }});
        prog.back().out = latents_out ? latents_out : moments_out; prog.back().out_elems = npix * (latents_out ? 4 : 8);
    }
    return HD_OK;
}

static int build_vae_decode(hd_ctx* c, int B, int L, const float* latents, float* images_out) {
    auto& prog = c->vae_dec_prog; prog.clear();
    auto& v = c->vws; auto& w = c->vw;
    int H = L;
    {
        uint4* in8 = v.in8; const float *pw = w.pq_w, *pb = w.pq_b; const int HW = H * H; const size_t npix = (size_t)B * HW;
        prog.push_back({"post_quant_conv", [=](hipStream_t s) -> hipError_t {
                            hipLaunchKernelGGL(vae_decode_entry_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, latents, pw, pb, in8, HW, npix, 1.0f / 0.18215f);
                            return hipGetLastError();
                        }});
    }
    vae_conv3(c, prog, "decoder.conv_in", w.dec_in, reinterpret_cast<const unsigned short*>(v.in8), B, H, 1, v.X, nullptr, nullptr);
    vae_resnet(c, prog, w.dec_mid[0], B, H, nullptr);
    vae_attention(c, prog, w.dec_attn, B, H);
    vae_resnet(c, prog, w.dec_mid[1], B, H, nullptr);
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 3; ++j) vae_resnet(c, prog, w.dec_res[i][j], B, H, nullptr);
        if (i < 3) {                                       // Upsample2D: nearest 2x, then conv 3x3
            const float* x = v.X; unsigned short* u = v.U; const int C = w.dec_res[i][2].cout, Hc = H;
            prog.push_back({"decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.nearest", [=](hipStream_t s) -> hipError_t {
                                const size_t n = (size_t)B * 4 * Hc * Hc * (C / 8);
                                hipLaunchKernelGGL(upsample2x_bf16_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 65535)), dim3(256), 0, s, x, u, B, Hc, Hc, C);
                                return hipGetLastError();
                            }});
            H *= 2;
            vae_conv3(c, prog, "decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv", w.dec_up[i], v.U, B, H, 1, v.X, nullptr, nullptr);
        }
    }
    vae_groupnorm(c, prog, "decoder.conv_norm_out", v.X, w.dec_nw, w.dec_nb, v.H, B, H, 128, true);
    vae_conv3(c, prog, "decoder.conv_out", w.dec_out, v.H, B, H, 1, v.out3, nullptr, nullptr);
    {
        const float* o3 = v.out3; const int HW = H * H; const size_t npix = (size_t)B * HW;
        prog.push_back({"decoder.output", [=](hipStream_t s) -> hipError_t {
                            hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((npix * 3 + 255) / 256)), dim3(256), 0, s, o3, images_out, 3, 3, HW, npix);
                            return hipGetLastError();
                        }});
        prog.back().out = images_out; prog.back().out_elems = npix * 3;
    }
    return HD_OK;
}

// XCD-local persistent stages (hd_xcd.hpp): usable when every one of the 256 workgroups gets a CU of its own (8 XCDs x 32
// CUs) and the level geometry is the latent-16 one (4 / 16 pixels per face at levels 3 / 2).  HD_NO_XCD=1 builds the
// program without them.
int setup_xcd(hd_ctx* c) {
    c->xcd_ok = false; c->face_ok = false;
    if (c->S != 1 || getenv("HD_NO_XCD")) return HD_OK;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) != hipSuccess || prop.multiProcessorCount != XS_GROUPS * XS_GROUP_WG) return HD_OK;
    HIPCHECK(c, hipHostMalloc(reinterpret_cast<void**>(&c->xcd_tmo_host), 64, hipHostMallocMapped));
    c->xcd_tmo_host[0] = 0;
    HIPCHECK(c, hipHostGetDevicePointer(reinterpret_cast<void**>(&c->xcd_tmo_dev), c->xcd_tmo_host, 0));
    if (int rc = dev_alloc(c, &c->abort_dev, 64)) return rc;
    HIPCHECK(c, hipMemset(c->abort_dev, 0, 64 * sizeof(unsigned)));
    c->xcd_ok = true;
    c->xcd2_mask = getenv("HD_XCD2") ? (atoi(getenv("HD_XCD2")) & 3) : 1;
    c->face_ok = getenv("HD_NO_FACE") == nullptr;
    if (const char* e = getenv("HD_FACE_L1_ROWS")) c->face_l1_rows = atoi(e) == 32 ? 32 : 16;        // per context, like xcd_ok: not a process-wide static (fixtures toggle the variable around make_model)
    return HD_OK;
}
// the device-side description of a stage (weights only: shared by every workspace), created on first use
int get_xstage(hd_ctx* c, int first_block, int nblocks, hd_ctx::XStage** out) {
    auto it = c->xstages.find(first_block);
    if (it != c->xstages.end() && it->second.nblocks == nblocks) { *out = &it->second; return HD_OK; }
    hd_ctx::XStage st;
    st.nblocks = nblocks;
    std::vector<XBlockW> host((size_t)nblocks);
    for (int j = 0; j < nblocks; ++j) {
        const BlockW& bw = c->den_blocks[first_block + j];
        XBlockW& x = host[j];
        x.w1 = bw.conv1.w; x.wsca = bw.sca.w; x.w3 = bw.conv3.w; x.w4 = bw.conv4.w; x.w5 = bw.conv5.w;
        x.b1 = bw.conv1.bias; x.bsca = bw.sca.bias; x.b3 = bw.conv3.bias; x.b4 = bw.conv4.bias; x.b5 = bw.conv5.bias;
        x.beta = bw.beta; x.gamma = bw.gamma; x.dw_w = bw.dw_wT; x.dw_b = bw.dw_b; x.film_off = bw.film_off; x.pad_ = 0;
    }
    const bool ws = c->ws_scope;
    c->ws_scope = false;                                  // context-lifetime allocations
    int rc = dev_alloc(c, &st.blocks_dev, (size_t)nblocks);
    rc |= dev_alloc(c, &st.sync, (size_t)3 * 256);
    c->ws_scope = ws;
    if (rc) return rc;
    HIPCHECK(c, hipMemcpy(st.blocks_dev, host.data(), host.size() * sizeof(XBlockW), hipMemcpyHostToDevice));
    HIPCHECK(c, hipMemset(st.sync, 0, (size_t)3 * 256 * sizeof(unsigned)));
    const int C = c->den_blocks[first_block].C, HW = (C == 1024) ? 4 : 16;
    if (c->xcd2_mask & (C == 1024 ? 2 : 1)) {
        // the five 1x1 convs of every block once more, in the A-operand order of the 16x16x32 MFMA (hd_xcd2.hpp)
        c->ws_scope = false;
        for (int j = 0; j < nblocks; ++j) {
            const BlockW& bw = c->den_blocks[first_block + j];
            const char* conv[5] = {".conv1", ".sca.1", ".conv3", ".conv4", ".conv5"};
            const uint4** dst[5] = {&host[j].w1, &host[j].wsca, &host[j].w3, &host[j].w4, &host[j].w5};
            for (int k = 0; k < 5; ++k) {
                const RawTensor* w = find_raw(c, bw.name + conv[k] + ".weight");
                if (!w || (int)w->shape[1] != C) { c->ws_scope = ws; HD_FAIL(c, HD_ERR_WEIGHTS, "missing %s%s.weight", bw.name.c_str(), conv[k]); }
                const int N = (int)w->shape[0];
                uint4* d = nullptr;
                if (int r2 = dev_alloc(c, &d, (size_t)N * C / 8)) { c->ws_scope = ws; return r2; }
                hipLaunchKernelGGL(pack_weight16_kernel, dim3(1024), dim3(256), 0, 0, w->dev, d, N, C);
                *dst[k] = d;
            }
        }
        int r2 = dev_alloc(c, &st.blocks2_dev, (size_t)nblocks);
        r2 |= dev_alloc(c, &st.sync2, (size_t)2048);
        const size_t hn = (size_t)64 * HW * C / 8;
        r2 |= dev_alloc(c, &st.hX, hn); r2 |= dev_alloc(c, &st.hG, hn); r2 |= dev_alloc(c, &st.hY, hn);
        r2 |= dev_alloc(c, &st.hsx, (size_t)64 * HW * (C / 16)); r2 |= dev_alloc(c, &st.hsy, (size_t)64 * HW * (C / 16));
        c->ws_scope = ws;
        if (r2) return r2;
        HIPCHECK(c, hipGetLastError());
        HIPCHECK(c, hipMemcpy(st.blocks2_dev, host.data(), host.size() * sizeof(XBlockW), hipMemcpyHostToDevice));
        HIPCHECK(c, hipMemset(st.sync2, 0, (size_t)2048 * sizeof(unsigned)));
        HIPCHECK(c, hipMemset(st.hX, 0, hn * 16)); HIPCHECK(c, hipMemset(st.hG, 0, hn * 16)); HIPCHECK(c, hipMemset(st.hY, 0, hn * 16));
        HIPCHECK(c, hipMemset(st.hsx, 0, (size_t)64 * HW * (C / 16) * 8)); HIPCHECK(c, hipMemset(st.hsy, 0, (size_t)64 * HW * (C / 16) * 8));
    }
    c->xstages[first_block] = st;
    *out = &c->xstages[first_block];
    return HD_OK;
}
// A hand-off wait of a persistent stage gave up (a workgroup was not resident, or a fault).  The call in which that
// happened hands back NaN (poison_if_abort_kernel at its end; the remaining stage launches of that call step aside at
// entry), and the failure is reported by the first check that runs after it -- hd_check() behind the caller's
// synchronisation, or the next hd_eps / hd_sample.  Reported once; the context then runs one launch per GEMM.
static int check_xcd(hd_ctx* c) {
    if (c->xcd_tmo_host && c->xcd_tmo_host[0]) {
        const unsigned code = c->xcd_tmo_host[0];
        (void)hipDeviceSynchronize();                      // the failed call's remaining launches still read the words reset below
        c->xcd_tmo_host[0] = 0;
        if (c->abort_dev) (void)hipMemset(c->abort_dev, 0, 64 * sizeof(unsigned));
        c->xcd_on = false; c->graphs_valid = false;
        for (auto& kv : c->ws_cache) kv.second.graphs_valid = false;
        for (auto& kv : c->xstages) {
            (void)hipMemset(kv.second.sync, 0, (size_t)3 * 256 * sizeof(unsigned));
            if (kv.second.sync2) (void)hipMemset(kv.second.sync2, 0, (size_t)2048 * sizeof(unsigned));
        }
        for (auto& kv : c->fstages) (void)hipMemset(kv.second.sync, 0, (size_t)2 * 64 * 16 * sizeof(unsigned));
        c->face_on = false;
        HD_FAIL(c, HD_ERR_HIP, "persistent stage: a hand-off wait gave up (code 0x%x%s); the results of that call are invalid (NaN), "
                               "the context now runs one launch per GEMM", code, c->stage_test_abort ? ", injected by stage_test_abort" : "");
    }
    return HD_OK;
}
// End of hd_eps / hd_sample: NaN into the call's result buffer when a stage gave up during the call.
static int poison_on_abort(hd_ctx* c, float* buf, size_t n, hipStream_t s) {
    if (!c->abort_dev || !c->xcd_ok) return HD_OK;
    hipLaunchKernelGGL(poison_if_abort_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, c->abort_dev, buf, n);
    HIPCHECK(c, hipGetLastError());
    return HD_OK;
}

}  // namespace

// ================================================================================================ C-ABI
extern "C" {

const char* hd_last_error(const hd_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int hd_create_unconditional(hd_ctx** out, int latent_res, int device) {
    int rc = hd_create(out, latent_res, device);
    if (rc == HD_OK) (*out)->conditional = false;
    return rc;
}

int hd_cr_create(hd_ctx** out, int device) {
    int rc = hd_create(out, 16, device);
    if (rc == HD_OK) { (*out)->cr = true; (*out)->conditional = false; }
    return rc;
}

// cr_face = CoarseRestoration(ln_face) (test_refiner.py:77): [B,3,128,128] fp32 NCHW in and out.
int hd_cr_forward(hd_ctx* c, int batch, const float* ln_face, float* cr_face_out, void* stream) {
    if (!c) return HD_ERR_INVALID;
    if (!c->cr) HD_FAIL(c, HD_ERR_INVALID, "hd_cr_forward: not a CoarseRestoration context (hd_cr_create)");
    if (!c->finalized) HD_FAIL(c, HD_ERR_NOT_READY, "weights are not loaded/finalized");
    if (!ln_face || !cr_face_out || batch <= 0 || batch > 1024) HD_FAIL(c, HD_ERR_INVALID, "hd_cr_forward: bad arguments");
    HIPCHECK(c, hipSetDevice(c->device));
    int rc = alloc_cr(c, batch);
    if (rc) return rc;
    if (c->cr_program.empty() || c->cr_in != ln_face || c->cr_out != cr_face_out) {
        rc = build_cr_program(c, ln_face, cr_face_out);
        if (rc) return rc;
    }
    return run_ops(c, c->cr_program, reinterpret_cast<hipStream_t>(stream), c->op_limit);
}

// AutoencoderKL boundary.  encode: images [B,3,in_res,in_res] fp32 NCHW -> bicubic to image_res (test_refiner.py:80)
// -> encoder -> quant_conv; moments_out [B,8,L,L] (mean | logvar) and / or latents_out [B,4,L,L] =
// latent_dist.sample() * 0.18215 (noise [B,4,L,L] or device Philox(seed)).  vae_range 1: clamp(0,1)*2-1 first
// (train_refiner.py:72-83).  decode: latents [B,4,L,L] -> decode(latents / 0.18215).sample [B,3,8L,8L] (test_refiner.py:93).
int hd_vae_create(hd_ctx** out, int device) {
    int rc = hd_create(out, 16, device);
    if (rc == HD_OK) { (*out)->vae = true; (*out)->conditional = false; }
    return rc;
}
int hd_vae_encode(hd_ctx* c, int batch, int in_res, int image_res, const float* images, int vae_range, const float* noise, uint64_t seed,
                  float* moments_out, float* latents_out, void* stream) {
    if (!c) return HD_ERR_INVALID;
    if (!c->vae) HD_FAIL(c, HD_ERR_INVALID, "hd_vae_encode: not an AutoencoderKL context (hd_vae_create)");
    if (!c->finalized) HD_FAIL(c, HD_ERR_NOT_READY, "weights are not loaded/finalized");
    if (!images || (!moments_out && !latents_out) || batch <= 0 || batch > 1024 || in_res < 8 || image_res < 64 || image_res > 512 || image_res % 64)
        HD_FAIL(c, HD_ERR_INVALID, "hd_vae_encode: bad arguments (image_res must be a multiple of 64 in [64, 512])");
    HIPCHECK(c, hipSetDevice(c->device));
    int rc = alloc_vae(c, batch, image_res);
    if (rc) return rc;
    const void* key[4] = {images, noise, moments_out, latents_out};
    const int flags = in_res * 4 + vae_range * 2;
    if (c->vae_enc_prog.empty() || memcmp(key, c->vae_enc_key, sizeof(key)) != 0 || flags != c->vae_enc_flags || seed != c->vae_seed) {
        rc = build_vae_encode(c, batch, in_res, image_res, images, vae_range, noise, seed, moments_out, latents_out);
        if (rc) return rc;
        memcpy(c->vae_enc_key, key, sizeof(key)); c->vae_enc_flags = flags;
    }
    c->vae_seed = seed;
    return run_ops(c, c->vae_enc_prog, reinterpret_cast<hipStream_t>(stream), c->op_limit);
}
int hd_vae_decode(hd_ctx* c, int batch, int latent_res, const float* latents, float* images_out, void* stream) {
    if (!c) return HD_ERR_INVALID;
    if (!c->vae) HD_FAIL(c, HD_ERR_INVALID, "hd_vae_decode: not an AutoencoderKL context (hd_vae_create)");
    if (!c->finalized) HD_FAIL(c, HD_ERR_NOT_READY, "weights are not loaded/finalized");
    if (!latents || !images_out || batch <= 0 || batch > 1024 || latent_res < 8 || latent_res > 64 || latent_res % 8) HD_FAIL(c, HD_ERR_INVALID, "hd_vae_decode: bad arguments");
    HIPCHECK(c, hipSetDevice(c->device));
    int rc = alloc_vae(c, batch, latent_res * 8);
    if (rc) return rc;
    const void* key[2] = {latents, images_out};
    if (c->vae_dec_prog.empty() || memcmp(key, c->vae_dec_key, sizeof(key)) != 0) {
        rc = build_vae_decode(c, batch, latent_res, latents, images_out);
        if (rc) return rc;
        memcpy(c->vae_dec_key, key, sizeof(key));
    }
    return run_ops(c, c->vae_dec_prog, reinterpret_cast<hipStream_t>(stream), c->op_limit);
}

int hd_create(hd_ctx** out, int latent_res, int device) {
    if (!out) return HD_ERR_INVALID;
    *out = nullptr;
    if (latent_res < 16 || latent_res % 16 != 0 || latent_res > 64) { g_create_error = "latent_res must be 16, 32, 48 or 64"; return HD_ERR_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_create_error = "no HIP device available"; return HD_ERR_HIP; }
    if (device < 0 || device >= ndev) { g_create_error = "device index out of range"; return HD_ERR_INVALID; }
    if (hipSetDevice(device) != hipSuccess) { g_create_error = "hipSetDevice failed"; return HD_ERR_HIP; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { g_create_error = "hipGetDeviceProperties failed"; return HD_ERR_HIP; }
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos) {
        g_create_error = std::string("this library is built for gfx950 (MI355X) only; device is ") + prop.gcnArchName;
        return HD_ERR_INVALID;
    }
    hd_ctx* c = new hd_ctx();
    c->L = latent_res; c->device = device; c->S = latent_res / 16;
    int rc = dev_alloc(c, &c->freq_dev, 64);
    if (rc) { g_create_error = c->err; hd_destroy(c); return rc; }
    float freq[64];
    const float e = (float)(-(std::log(10000.0) / 63.0));       // model.py:25: python double, then fp32 tensor math
    for (int k = 0; k < 64; ++k) freq[k] = expf((float)k * e);
    (void)hipMemcpy(c->freq_dev, freq, sizeof(freq), hipMemcpyHostToDevice);
    (void)hipEventCreate(&c->ev0); (void)hipEventCreate(&c->ev1);
    (void)hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming);
    *out = c;
    return HD_OK;
}

void hd_destroy(hd_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (auto& ch : c->chains) {
        if (ch.graph_exec) (void)hipGraphExecDestroy(ch.graph_exec);
        if (ch.graph_multi) (void)hipGraphExecDestroy(ch.graph_multi);
        if (ch.stream) (void)hipStreamDestroy(ch.stream);
        if (ch.done) (void)hipEventDestroy(ch.done);
    }
    if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
    if (c->film_ev) (void)hipEventDestroy(c->film_ev);
    for (auto& sg : c->stage) { if (sg.ev) (void)hipEventDestroy(sg.ev); if (sg.host) (void)hipHostFree(sg.host); }
    if (c->xcd_tmo_host) (void)hipHostFree(c->xcd_tmo_host);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (auto& kv : c->ws_cache) destroy_saved(kv.second);
    for (auto& kv : c->raw) if (kv.second.dev) (void)hipFree(kv.second.dev);
    for (void* p : c->allocs) if (p) (void)hipFree(p);
    for (void* p : c->ws_allocs) if (p) (void)hipFree(p);
    delete c;
}

int hd_load_weights(hd_ctx* c, const hd_tensor_desc* t, int n) {
    if (!c || (!t && n > 0)) return HD_ERR_INVALID;
    if (c->finalized) HD_FAIL(c, HD_ERR_INVALID, "weights already finalized");
    HIPCHECK(c, hipSetDevice(c->device));
    for (int i = 0; i < n; ++i) {
        if (!t[i].name || t[i].ndim < 0 || t[i].ndim > 4) HD_FAIL(c, HD_ERR_INVALID, "bad tensor descriptor %d", i);
        const std::string name = t[i].name;
        RawTensor r;
        r.numel = 1;
        for (int d = 0; d < t[i].ndim; ++d) { r.shape.push_back(t[i].shape[d]); r.numel *= (size_t)t[i].shape[d]; }
        const bool counter = name.size() > 19 && name.compare(name.size() - 19, 19, "num_batches_tracked") == 0;
        if (!counter) {
            if (!t[i].data) HD_FAIL(c, HD_ERR_INVALID, "tensor %s has no data", name.c_str());
            HIPCHECK(c, hipMalloc(reinterpret_cast<void**>(&r.dev), r.numel * sizeof(float) + 256));
            HIPCHECK(c, hipMemcpy(r.dev, t[i].data, r.numel * sizeof(float), t[i].is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
            if (r.numel <= HOST_MIRROR_MAX) {
                r.host.resize(r.numel);
                HIPCHECK(c, hipMemcpy(r.host.data(), t[i].data, r.numel * sizeof(float), t[i].is_device ? hipMemcpyDeviceToHost : hipMemcpyHostToHost));
            }
        }
        auto it = c->raw.find(name);
        if (it != c->raw.end() && it->second.dev) (void)hipFree(it->second.dev);
        c->raw[name] = std::move(r);
    }
    return HD_OK;
}

int hd_finalize_weights(hd_ctx* c) {
    if (!c) return HD_ERR_INVALID;
    if (c->finalized) return HD_OK;
    HIPCHECK(c, hipSetDevice(c->device));
    if (c->cr) return finalize_cr(c);
    if (c->vae) return finalize_vae(c);
    // ---- strict key / shape check ----
    const auto man = build_manifest(c->L, c->conditional);
    for (const auto& e : man) {
        const RawTensor* r = find_raw(c, e.first);
        if (!r) HD_FAIL(c, HD_ERR_WEIGHTS, "Missing key in state_dict: %s", e.first.c_str());
        if (r->shape != e.second) HD_FAIL(c, HD_ERR_WEIGHTS, "size mismatch for %s", e.first.c_str());
    }
    if (c->raw.size() != man.size()) {
        std::unordered_map<std::string, int> known;
        for (const auto& e : man) known[e.first] = 1;
        for (const auto& kv : c->raw)
            if (!known.count(kv.first)) HD_FAIL(c, HD_ERR_WEIGHTS, "Unexpected key in state_dict: %s", kv.first.c_str());
    }
    int rc = 0;
    // ---- NAF blocks (denoiser in execution order, then FPG) ----
    auto load_block = [&](const std::string& p, int C, BlockW& bw) -> int {
        bw.name = p; bw.C = C;
        int r = 0;
        r |= pack_weight(c, p + ".conv1", &bw.conv1); r |= pack_weight(c, p + ".conv3", &bw.conv3);
        r |= pack_weight(c, p + ".sca.1", &bw.sca); r |= pack_weight(c, p + ".conv4", &bw.conv4);
        r |= pack_weight(c, p + ".conv5", &bw.conv5);
        if (r) return r;
        bw.dw_w = find_raw(c, p + ".conv2.weight")->dev; bw.dw_b = find_raw(c, p + ".conv2.bias")->dev;
        bw.beta = find_raw(c, p + ".beta")->dev; bw.gamma = find_raw(c, p + ".gamma")->dev;
        return make_dw_layout(c, bw);
    };
    const int enc[4] = {2, 2, 4, 8};
    std::vector<std::pair<std::string, int>> order;
    for (int l = 0; l < 4; ++l) for (int j = 0; j < enc[l]; ++j) order.push_back({"denoiser.encoders." + std::to_string(l) + "." + std::to_string(j), WIDTH << l});
    for (int j = 0; j < 8; ++j) order.push_back({"denoiser.middle_blks." + std::to_string(j), WIDTH << 4});
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) order.push_back({"denoiser.decoders." + std::to_string(i) + "." + std::to_string(j), WIDTH << (3 - i)});
    int off = 0;
    for (auto& e : order) {
        BlockW bw;
        rc = load_block(e.first, e.second, bw);
        if (rc) return rc;
        bw.film_off = off;
        off += 4 * e.second;
        c->den_block_index[e.first] = (int)c->den_blocks.size();
        c->den_blocks.push_back(bw);
    }
    c->film_total = off;
    rc = setup_xcd(c);
    if (rc) return rc;
    off = 0;
    for (int l = 0; l < 4 && c->conditional; ++l)
        for (int j = 0; j < enc[l]; ++j) {
            BlockW bw;
            rc = load_block("fpg.encoders." + std::to_string(l) + "." + std::to_string(j), WIDTH << l, bw);
            if (rc) return rc;
            bw.film_off = off;
            off += 4 * (WIDTH << l);
            c->fpg_blocks.push_back(bw);
        }
    const int fpg_film_total = off;
    // ---- FiLM: concatenated Linear(256,4C) weights/biases, LN affine in table layout ----
    rc |= dev_alloc(c, &c->film_W, (size_t)c->film_total * FILM_IN); rc |= dev_alloc(c, &c->film_b, c->film_total);
    rc |= dev_alloc(c, &c->ln_pack, c->film_total); rc |= dev_alloc(c, &c->fpg_ln_pack, fpg_film_total + 4);
    rc |= dev_alloc(c, &c->film_blocks_dev, c->den_blocks.size());
    if (rc) return rc;
    std::vector<FilmBlock> fbs;
    auto copy_ln = [&](const BlockW& bw, float* dst) -> int {
        const char* names[4] = {".norm1.bias", ".norm1.weight", ".norm2.bias", ".norm2.weight"};
        for (int q = 0; q < 4; ++q)
            HIPCHECK(c, hipMemcpy(dst + bw.film_off + q * bw.C, find_raw(c, bw.name + names[q])->dev, bw.C * sizeof(float), hipMemcpyDeviceToDevice));
        return HD_OK;
    };
    for (const BlockW& bw : c->den_blocks) {
        HIPCHECK(c, hipMemcpy(c->film_W + (size_t)bw.film_off * FILM_IN, find_raw(c, bw.name + ".mlp.1.weight")->dev, (size_t)4 * bw.C * FILM_IN * sizeof(float), hipMemcpyDeviceToDevice));
        HIPCHECK(c, hipMemcpy(c->film_b + bw.film_off, find_raw(c, bw.name + ".mlp.1.bias")->dev, (size_t)4 * bw.C * sizeof(float), hipMemcpyDeviceToDevice));
        rc = copy_ln(bw, c->ln_pack);
        if (rc) return rc;
        fbs.push_back({bw.film_off, bw.C});
    }
    for (const BlockW& bw : c->fpg_blocks) { rc = copy_ln(bw, c->fpg_ln_pack); if (rc) return rc; }
    HIPCHECK(c, hipMemcpy(c->film_blocks_dev, fbs.data(), fbs.size() * sizeof(FilmBlock), hipMemcpyHostToDevice));
    // ---- intro / ending weights re-laid for coalesced per-lane loads ----
    {
        const RawTensor *iw = find_raw(c, "denoiser.intro.weight"), *ew = find_raw(c, "denoiser.ending.weight");
        const RawTensor* fw = c->conditional ? find_raw(c, "fpg.intro.weight") : iw;
        if (!iw || !fw || !ew) HD_FAIL(c, HD_ERR_INVALID, "intro/ending weights missing");
        rc |= dev_alloc(c, &c->intro_wT, 36 * 128); rc |= dev_alloc(c, &c->fpg_intro_wT, 36 * 128); rc |= dev_alloc(c, &c->ending_wT, 9 * 4 * 128);
        if (rc) return rc;
        hipLaunchKernelGGL(intro_weight_layout_kernel, dim3(18), dim3(256), 0, 0, iw->dev, c->intro_wT);
        hipLaunchKernelGGL(intro_weight_layout_kernel, dim3(18), dim3(256), 0, 0, fw->dev, c->fpg_intro_wT);
        hipLaunchKernelGGL(ending_weight_layout_kernel, dim3(18), dim3(256), 0, 0, ew->dev, c->ending_wT);
        HIPCHECK(c, hipGetLastError());
    }
    // ---- downs / ups / fpg convs / idc_conv ----
    for (int i = 0; i < 4; ++i) {
        rc |= pack_weight(c, "denoiser.downs." + std::to_string(i), &c->den_down[i]);
        { PackOpts o; o.S2 = 4; rc |= pack_weight(c, "denoiser.ups." + std::to_string(i) + ".0", &c->den_up[i], o); }   // sub-pixel major (EpPixShufF32)
        if (c->conditional) rc |= pack_weight(c, "fpg.downs." + std::to_string(i), &c->fpg_down[i]);
    }
    for (int i = 0; i < 5 && c->conditional; ++i) { PackOpts o; o.S2 = i ? 4 : 1; rc |= pack_weight(c, "fpg.convs." + std::to_string(i) + ".0", &c->fpg_convs[i], o); }
    if (c->conditional) { PackOpts o; o.S2 = c->S * c->S; rc |= pack_weight(c, "denoiser.idc_conv", &c->idc_conv, o); }
    if (rc) return rc;
    // ---- HCAs ----
    for (int i = 0; i < 5 && c->conditional; ++i) {
        HcaW& h = c->hca[i];
        const std::string p = "denoiser.hcas." + std::to_string(i);
        h.C = (WIDTH << 4) >> i;
        h.centre_only = ((c->L >> (4 - i)) == 1);         // 1x1 map: only the centre tap sees data
        rc |= pack_weight(c, p + ".channel_mlp.0", &h.mlp0); rc |= pack_weight(c, p + ".channel_mlp.2", &h.mlp2);
        { PackOpts o; o.bn = p + ".spatial_mlp.1"; rc |= pack_weight(c, p + ".spatial_mlp.0", &h.sp0, o); }
        { PackOpts o; o.bn = p + ".fused_mlp.1"; o.centre_only = h.centre_only; rc |= pack_weight(c, p + ".fused_mlp.0", &h.fused, o); }
        if (rc) return rc;
        std::vector<float> s, o2;
        rc = bn_affine(c, p + ".spatial_mlp.4", s, o2);
        if (rc) return rc;
        const RawTensor *w3 = find_raw(c, p + ".spatial_mlp.3.weight"), *b3 = find_raw(c, p + ".spatial_mlp.3.bias");
        std::vector<float> wf(w3->numel);
        for (size_t k = 0; k < w3->numel; ++k) wf[k] = w3->host[k] * s[0];
        h.sp3_b = b3->host[0] * s[0] + o2[0];
        rc = upload_vec(c, wf, &h.sp3_w);
        if (rc) return rc;
    }
    // ---- ResNet-50 (conv + BN folded) ----
    auto load_res = [&](const std::string& conv, const std::string& bn, int cin, int cout, int k, int stride, int pad, ResConv& r) -> int {
        PackOpts o; o.bn = bn;
        if (cin == 3) o.cin_pad = 8;
        r.cin = cin; r.cout = cout; r.k = k; r.stride = stride; r.pad = pad;
        return pack_weight(c, conv, &r.w, o);
    };
    if (c->conditional) rc = load_res("idc.conv1", "idc.batch_norm1", 3, 64, 7, 2, 3, c->res_conv1);
    if (rc) return rc;
    if (c->conditional) {
        const int res_layers[4] = {3, 4, 6, 3}, planes[4] = {64, 128, 256, 512};
        int cin = 64;
        for (int li = 0; li < 4; ++li)
            for (int b = 0; b < res_layers[li]; ++b) {
                const std::string q = "idc.layer" + std::to_string(li + 1) + "." + std::to_string(b);
                const int stride = (b == 0 && li > 0) ? 2 : 1;
                ResBlock rb;
                rc |= load_res(q + ".conv1", q + ".batch_norm1", cin, planes[li], 1, 1, 0, rb.c1);
                rc |= load_res(q + ".conv2", q + ".batch_norm2", planes[li], planes[li], 3, stride, 1, rb.c2);
                rc |= load_res(q + ".conv3", q + ".batch_norm3", planes[li], planes[li] * 4, 1, 1, 0, rb.c3);
                if (b == 0) { rb.has_ds = true; rc |= load_res(q + ".i_downsample.0", q + ".i_downsample.1", cin, planes[li] * 4, 1, stride, 0, rb.ds); }
                if (rc) return rc;
                c->res_blocks.push_back(rb);
                cin = planes[li] * 4;
            }
    }
    HIPCHECK(c, hipDeviceSynchronize());
    // ---- algorithmic per-step figures of the step-variant denoiser (effective taps only) ----
    int64_t params = 0;
    double macs = 0.0;   // per face
    auto add_w = [&](const PackedW& w, double rows_per_face) { params += (int64_t)w.N * w.K; macs += rows_per_face * (double)w.N * w.K; };
    {
        int bi = 0;
        auto blocks_at = [&](int l, int n) {
            const double hw = (double)(c->L >> l) * (c->L >> l);
            for (int j = 0; j < n; ++j) {
                const BlockW& b = c->den_blocks[bi++];
                add_w(b.conv1, hw); add_w(b.conv3, hw); add_w(b.conv4, hw); add_w(b.conv5, hw); add_w(b.sca, 1.0);
                params += 2 * b.C * 9 + 2 * b.C; macs += hw * 2 * b.C * 9;   // depthwise (nominal taps)
            }
        };
        for (int l = 0; l < 4; ++l) blocks_at(l, enc[l]);
        blocks_at(4, 8);
        for (int i = 0; i < 4; ++i) blocks_at(3 - i, 2);
        for (int l = 0; l < 4; ++l) {
            const double hwd = (double)(c->L >> (l + 1)) * (c->L >> (l + 1));
            add_w(c->den_down[l], hwd);
            add_w(c->den_up[l], (double)(c->L >> (4 - l)) * (c->L >> (4 - l)));
        }
        for (int i = 0; i < 5 && c->conditional; ++i) { const double hw = (double)(c->L >> (4 - i)) * (c->L >> (4 - i)); add_w(c->hca[i].fused, hw); }
        params += 2 * 128 * 36; macs += 2.0 * c->L * c->L * 128 * 36;
    }
    c->weight_bytes_per_step = params * 2;
    c->flops_per_face_step = 2.0 * macs;
    c->finalized = true;
    return HD_OK;
}

static int check_ready(hd_ctx* c, bool need_prepared) {
    if (!c) return HD_ERR_INVALID;
    if (!c->finalized) HD_FAIL(c, HD_ERR_NOT_READY, "weights are not loaded/finalized");
    if (need_prepared && !c->prepared) HD_FAIL(c, HD_ERR_NOT_READY, "hd_prepare has not been called for this batch");
    return HD_OK;
}

static int prepare_common(hd_ctx* c, int batch) {
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (batch <= 0 || batch > 4096) HD_FAIL(c, HD_ERR_INVALID, "batch must be in [1, 4096]");
    HIPCHECK(c, hipSetDevice(c->device));
    return alloc_workspace(c, batch);
}

#define HD_NEED_CONDITIONAL(c, what) \
    do { if ((c) && !(c)->conditional) HD_FAIL(c, HD_ERR_INVALID, what ": this context holds the unconditional Denoiser or CoarseRestoration (no priors / identity)"); } while (0)

// Unconditional Denoiser: nothing to condition on -- size the workspace for `batch` faces and build the launch program.
int hd_prepare_unconditional(hd_ctx* c, int batch, void* stream) {
    (void)stream;
    if (c && c->cr) HD_FAIL(c, HD_ERR_INVALID, "hd_prepare_unconditional: this context holds CoarseRestoration");
    if (c && c->conditional) HD_FAIL(c, HD_ERR_INVALID, "hd_prepare_unconditional: this context holds the conditional FusedDenoiser");
    int rc = prepare_common(c, batch);
    if (rc) return rc;
    c->prepared = true;
    return HD_OK;
}

int hd_prepare(hd_ctx* c, int batch, const float* cr_latent, const float* cr_face, const float* id_emb, void* stream) {
    HD_NEED_CONDITIONAL(c, "hd_prepare");
    int rc = prepare_common(c, batch);
    if (rc) return rc;
    if (!cr_latent || (!cr_face == !id_emb)) HD_FAIL(c, HD_ERR_INVALID, "need cr_latent and exactly one of cr_face / id_emb");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t lat_face = (size_t)4 * c->L * c->L;
    for (auto& ch : c->chains) {
        c->ch = &ch;
        std::vector<Op> prog;
        add_fpg(c, prog, cr_latent + ch.face0 * lat_face);
        if (cr_face) add_resnet(c, prog, cr_face + (size_t)ch.face0 * 3 * 128 * 128);
        else HIPCHECK(c, hipMemcpyAsync(ch.id_emb, id_emb + (size_t)ch.face0 * 2048, (size_t)ch.B * 2048 * sizeof(float), hipMemcpyDeviceToDevice, s));
        for (int i = 0; i < 5; ++i) add_gates(c, prog, i);
        add_idc_term(c, prog);
        rc = run_ops(c, prog, s, ch.index == 0 ? c->prep_limit : -1);
        ch.prep_program.swap(prog);
        if (rc) break;
    }
    c->ch = &c->chains[0];
    if (rc) return rc;
    c->prepared = true;
    return HD_OK;
}

int hd_prepare_from_priors(hd_ctx* c, int batch, const float* const priors[5], const float* id_emb, void* stream) {
    HD_NEED_CONDITIONAL(c, "hd_prepare_from_priors");
    int rc = prepare_common(c, batch);
    if (rc) return rc;
    if (!priors || !id_emb) HD_FAIL(c, HD_ERR_INVALID, "priors and id_emb are required");
    for (int i = 0; i < 5; ++i) if (!priors[i]) HD_FAIL(c, HD_ERR_INVALID, "prior %d is NULL", i);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    for (auto& ch : c->chains) {
        c->ch = &ch;
        for (int i = 0; i < 5; ++i) {
            const Level& lv = ch.lv[4 - i];
            const size_t total = (size_t)lv.M * lv.C;          // per chain; faces are contiguous in NCHW too
            hipLaunchKernelGGL(nchw_to_nhwc_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                               priors[i] + (size_t)ch.face0 * lv.C * lv.H * lv.H, ch.prior[i], lv.C, lv.H * lv.H, total);
        }
        HIPCHECK(c, hipGetLastError());
        HIPCHECK(c, hipMemcpyAsync(ch.id_emb, id_emb + (size_t)ch.face0 * 2048, (size_t)ch.B * 2048 * sizeof(float), hipMemcpyDeviceToDevice, s));
        std::vector<Op> prog;
        for (int i = 0; i < 5; ++i) add_gates(c, prog, i);
        add_idc_term(c, prog);
        rc = run_ops(c, prog, s);
        if (rc) break;
    }
    c->ch = &c->chains[0];
    if (rc) return rc;
    c->prepared = true;
    return HD_OK;
}

int hd_fpg(hd_ctx* c, int batch, const float* cr_latent, float* const priors_out[5], void* stream) {
    HD_NEED_CONDITIONAL(c, "hd_fpg");
    int rc = prepare_common(c, batch);
    if (rc) return rc;
    if (!cr_latent || !priors_out) HD_FAIL(c, HD_ERR_INVALID, "hd_fpg: bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t lat_face = (size_t)4 * c->L * c->L;
    for (auto& ch : c->chains) {
        c->ch = &ch;
        std::vector<Op> prog;
        add_fpg(c, prog, cr_latent + ch.face0 * lat_face);
        rc = run_ops(c, prog, s);
        if (rc) break;
        for (int i = 0; i < 5; ++i) {
            if (!priors_out[i]) continue;
            const Level& lv = ch.lv[4 - i];
            const size_t total = (size_t)lv.M * lv.C;
            hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, ch.prior[i],
                               priors_out[i] + (size_t)ch.face0 * lv.C * lv.H * lv.H, lv.C, lv.H * lv.H, total);
        }
    }
    c->ch = &c->chains[0];
    if (rc) return rc;
    HIPCHECK(c, hipGetLastError());
    return HD_OK;
}

int hd_idc(hd_ctx* c, int batch, const float* cr_face, float* id_emb_out, void* stream) {
    HD_NEED_CONDITIONAL(c, "hd_idc");
    int rc = prepare_common(c, batch);
    if (rc) return rc;
    if (!cr_face || !id_emb_out) HD_FAIL(c, HD_ERR_INVALID, "hd_idc: bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    for (auto& ch : c->chains) {
        c->ch = &ch;
        std::vector<Op> prog;
        add_resnet(c, prog, cr_face + (size_t)ch.face0 * 3 * 128 * 128);
        rc = run_ops(c, prog, s);
        if (rc) break;
        HIPCHECK(c, hipMemcpyAsync(id_emb_out + (size_t)ch.face0 * 2048, ch.id_emb, (size_t)ch.B * 2048 * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    c->ch = &c->chains[0];
    return rc;
}

int hd_scheduler_step(float* x_inout, const float* eps, const float* coef7, const float* noise, uint64_t seed, int step,
                      int64_t n_elems, void* stream) {
    if (!x_inout || !eps || !coef7 || n_elems <= 0) return HD_ERR_INVALID;
    Coef7 k;
    for (int i = 0; i < 7; ++i) k.c[i] = coef7[i];
    hipLaunchKernelGGL(sched_step_direct_kernel, dim3((unsigned)((n_elems + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       x_inout, eps, k, noise, (unsigned long long)seed, step, (long long)n_elems);
    return hipGetLastError() == hipSuccess ? HD_OK : HD_ERR_HIP;
}

int hd_eps(hd_ctx* c, const float* x, const float* timesteps, int n_t, float* eps_out, void* stream) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    rc = check_xcd(c);
    if (rc) return rc;
    if (!x || !timesteps || !eps_out || (n_t != 1 && n_t != c->B)) HD_FAIL(c, HD_ERR_INVALID, "hd_eps: bad arguments (n_t must be 1 or batch)");
    HIPCHECK(c, hipSetDevice(c->device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    rc = ensure_film_rows(c, n_t);
    if (rc) return rc;
    const size_t nlat = (size_t)c->B * 4 * c->L * c->L;
    HIPCHECK(c, hipMemcpyAsync(c->lat, x, nlat * sizeof(float), hipMemcpyDeviceToDevice, s));
    for (auto& ch : c->chains) HIPCHECK(c, hipMemsetAsync(ch.step_state, 0, sizeof(StepState), s));
    c->film_valid = false;                               // rows [0, n_t) are overwritten
    rc = compute_film(c, timesteps, n_t, s);
    if (rc) return rc;
    c->film_step_stride = 0;
    c->film_from_cur = false;
    c->film_face_stride = (n_t == 1) ? 0 : c->film_total;
    c->advance = 0;
    for (auto& ch : c->chains) {
        rc = run_ops(c, ch.program, s, ch.index == 0 ? c->op_limit : -1);
        if (rc) return rc;
    }
    rc = poison_on_abort(c, c->eps, nlat, s);
    if (rc) return rc;
    HIPCHECK(c, hipMemcpyAsync(eps_out, c->eps, nlat * sizeof(float), hipMemcpyDeviceToDevice, s));
    return HD_OK;
}

int hd_sample(hd_ctx* c, float* x_inout, const hd_schedule* sched, const float* noise, uint64_t seed, void* stream) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    rc = check_xcd(c);
    if (rc) return rc;
    if (!x_inout || !sched || sched->n_steps <= 0 || !sched->timesteps || !sched->coef) HD_FAIL(c, HD_ERR_INVALID, "hd_sample: bad arguments");
    HIPCHECK(c, hipSetDevice(c->device));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int n = sched->n_steps;
    rc = ensure_film_rows(c, n);
    if (rc) return rc;
    if (n > c->coef_cap) {
        dev_free(c, c->coef_dev);
        rc = dev_alloc(c, &c->coef_dev, (size_t)n * 7);
        if (rc) return rc;
        c->coef_cap = n;
        c->graphs_valid = false;
        // parked workspaces captured the old coefficient buffer into their ending launch as well (SchedArgs::coef)
        for (auto& kv : c->ws_cache) kv.second.graphs_valid = false;
    }
    const size_t per_face = (size_t)4 * c->L * c->L;
    const size_t nlat = (size_t)c->B * per_face;
    // schedule and loop state (step = -1: each chain's intro kernel pre-increments) go through a pinned staging buffer of
    // the context, so the caller's host arrays are free on return and nothing here waits for the stream
    StepState st{};
    st.step = -1; st.n_steps = n; st.noise = noise; st.seed = seed;
    {
        auto& sg = c->stage[c->stage_idx ^= 1];
        const size_t st_f = (sizeof(StepState) + 3) / 4, need = (size_t)n * 8 + st_f;
        if (sg.pending) { HIPCHECK(c, hipEventSynchronize(sg.ev)); sg.pending = false; }     // the copy issued two calls ago
        if (sg.cap < need) {
            if (sg.host) (void)hipHostFree(sg.host);
            sg.host = nullptr; sg.cap = 0;
            HIPCHECK(c, hipHostMalloc(reinterpret_cast<void**>(&sg.host), need * sizeof(float), hipHostMallocDefault));
            sg.cap = need;
        }
        if (!sg.ev) HIPCHECK(c, hipEventCreateWithFlags(&sg.ev, hipEventDisableTiming));
        memcpy(sg.host, sched->coef, (size_t)n * 7 * sizeof(float));
        memcpy(sg.host + (size_t)n * 7, sched->timesteps, (size_t)n * sizeof(float));
        memcpy(sg.host + (size_t)n * 8, &st, sizeof(st));
        HIPCHECK(c, hipMemcpyAsync(c->coef_dev, sg.host, (size_t)n * 7 * sizeof(float), hipMemcpyHostToDevice, s));
        for (auto& ch : c->chains) HIPCHECK(c, hipMemcpyAsync(ch.step_state, sg.host + (size_t)n * 8, sizeof(st), hipMemcpyHostToDevice, s));
        const bool same_sched = c->film_valid && c->film_sched.size() == (size_t)n &&
                                memcmp(c->film_sched.data(), sched->timesteps, (size_t)n * sizeof(float)) == 0;
        if (!same_sched) HIPCHECK(c, hipMemcpyAsync(c->t_dev, sg.host + (size_t)n * 7, (size_t)n * sizeof(float), hipMemcpyHostToDevice, s));
        HIPCHECK(c, hipEventRecord(sg.ev, s));
        sg.pending = true;
        HIPCHECK(c, hipMemcpyAsync(c->lat, x_inout, nlat * sizeof(float), hipMemcpyDeviceToDevice, s));
        if (!same_sched) {                                // FiLM table of the whole schedule (0.5 GB at 1000 steps): once per schedule
            c->film_valid = false;
            rc = compute_film(c, c->t_dev, n, s);
            if (rc) return rc;
            if (!c->film_ev) HIPCHECK(c, hipEventCreateWithFlags(&c->film_ev, hipEventDisableTiming));
            HIPCHECK(c, hipEventRecord(c->film_ev, s));
            c->film_sched.assign(sched->timesteps, sched->timesteps + n);
            c->film_valid = true;
        } else {
            HIPCHECK(c, hipStreamWaitEvent(s, c->film_ev, 0));       // no-op on the stream that computed it
        }
    }
    c->film_step_stride = c->film_total;
    c->film_face_stride = 0;
    c->film_from_cur = true;
    c->advance = 1;
    for (auto& ch : c->chains)                          // step 0's row; the ending launch of step i stages row i+1
        HIPCHECK(c, hipMemcpyAsync(ch.film_cur, c->film_table, (size_t)c->film_total * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (!c->graphs_valid || c->graph_film != c->film_table || c->graph_B != c->B) {
        // One graph per chain: its launch program + its scheduler update.  Faces never interact, so the
        // chains are independent over the whole loop and each graph is replayed on the chain's own stream.
        for (auto& ch : c->chains) {
            if (ch.graph_exec) { (void)hipGraphExecDestroy(ch.graph_exec); ch.graph_exec = nullptr; }
            if (ch.graph_multi) { (void)hipGraphExecDestroy(ch.graph_multi); ch.graph_multi = nullptr; }
            for (int multi = 0; multi < 2; ++multi) {       // one step, and kGraphSteps steps back to back (fewer graph launches)
                hipGraph_t graph = nullptr;
                hipError_t e = hipStreamBeginCapture(ch.stream, hipStreamCaptureModeThreadLocal);
                if (e == hipSuccess) {
                    for (int r = 0; r < (multi ? kGraphSteps : 1) && e == hipSuccess; ++r)
                        for (size_t k = 0; k < ch.program.size() && e == hipSuccess; ++k) e = ch.program[k].run(ch.stream);
                    hipError_t e2 = hipStreamEndCapture(ch.stream, &graph);
                    if (e == hipSuccess) e = e2;
                }
                if (e == hipSuccess) e = hipGraphInstantiate(multi ? &ch.graph_multi : &ch.graph_exec, graph, nullptr, nullptr, 0);
                if (graph) (void)hipGraphDestroy(graph);
                if (e != hipSuccess) HD_FAIL(c, HD_ERR_HIP, "graph capture/instantiate failed: %s", hipGetErrorString(e));
            }
        }
        c->graphs_valid = true; c->graph_film = c->film_table; c->graph_B = c->B;
    }
    if (c->profiling) HIPCHECK(c, hipEventRecord(c->ev0, s));
    HIPCHECK(c, hipEventRecord(c->fork_ev, s));
    for (auto& ch : c->chains) HIPCHECK(c, hipStreamWaitEvent(ch.stream, c->fork_ev, 0));
    {
        int i = 0;
        for (; i + kGraphSteps <= n; i += kGraphSteps)
            for (auto& ch : c->chains) HIPCHECK(c, hipGraphLaunch(ch.graph_multi, ch.stream));
        for (; i < n; ++i)
            for (auto& ch : c->chains) HIPCHECK(c, hipGraphLaunch(ch.graph_exec, ch.stream));
    }
    for (auto& ch : c->chains) {
        HIPCHECK(c, hipEventRecord(ch.done, ch.stream));
        HIPCHECK(c, hipStreamWaitEvent(s, ch.done, 0));
    }
    if (c->profiling) { HIPCHECK(c, hipEventRecord(c->ev1, s)); c->last_steps = n; }
    rc = poison_on_abort(c, c->lat, nlat, s);
    if (rc) return rc;
    HIPCHECK(c, hipMemcpyAsync(x_inout, c->lat, nlat * sizeof(float), hipMemcpyDeviceToDevice, s));
    return HD_OK;
}

static std::vector<Op>* which_program(hd_ctx* c, int which) {
    static std::vector<Op> empty;
    if (c->cr) return &c->cr_program;
    if (c->vae) return which == 0 ? &c->vae_enc_prog : &c->vae_dec_prog;
    if (c->chains.empty()) return &empty;
    return which == 0 ? &c->chains[0].program : &c->chains[0].prep_program;
}
int hd_num_ops(hd_ctx* c, int which) { return c ? (int)which_program(c, which)->size() : 0; }
int hd_num_chains(hd_ctx* c) { return c ? (int)c->chains.size() : 0; }
int hd_debug_limit_ops(hd_ctx* c, int which, int n) {
    if (!c) return HD_ERR_INVALID;
    (which == 0 ? c->op_limit : c->prep_limit) = n;
    return HD_OK;
}
const char* hd_debug_op_name(hd_ctx* c, int which, int i) {
    if (!c) return "";
    auto* p = which_program(c, which);
    return (i >= 0 && i < (int)p->size()) ? (*p)[i].name.c_str() : "";
}
static int64_t read_to_host(hd_ctx* c, const void* dev, size_t n, int is_bf16, float* host_out, int64_t max_elems) {
    if (!host_out) return (int64_t)n;
    if ((int64_t)n > max_elems) HD_FAIL(c, HD_ERR_INVALID, "debug read needs %zu elements", n);
    HIPCHECK(c, hipSetDevice(c->device));
    HIPCHECK(c, hipDeviceSynchronize());
    if (is_bf16) {
        std::vector<unsigned short> tmp(n);
        HIPCHECK(c, hipMemcpy(tmp.data(), dev, n * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) { unsigned u = (unsigned)tmp[i] << 16; memcpy(&host_out[i], &u, 4); }
    } else {
        HIPCHECK(c, hipMemcpy(host_out, dev, n * sizeof(float), hipMemcpyDeviceToHost));
    }
    return (int64_t)n;
}
int64_t hd_debug_read_op(hd_ctx* c, int which, int i, float* host_out, int64_t max_elems) {
    if (!c) return HD_ERR_INVALID;
    auto* p = which_program(c, which);
    if (i < 0 || i >= (int)p->size() || !(*p)[i].out) HD_FAIL(c, HD_ERR_INVALID, "no such op %d", i);
    return read_to_host(c, (*p)[i].out, (*p)[i].out_elems, (*p)[i].out_bf16, host_out, max_elems);
}

int64_t hd_debug_read(hd_ctx* c, const char* name, float* host_out, int64_t max_elems) {
    if (!c || !name) return HD_ERR_INVALID;
    auto it = c->dbg.find(name);
    if (it == c->dbg.end()) HD_FAIL(c, HD_ERR_INVALID, "unknown debug buffer %s", name);
    return read_to_host(c, it->second.first, it->second.second.first, it->second.second.second, host_out, max_elems);
}

int hd_debug_write(hd_ctx* c, const char* name, const float* host_in, int64_t n_elems) {
    if (!c || !name || !host_in) return HD_ERR_INVALID;
    auto it = c->dbg.find(name);
    if (it == c->dbg.end()) HD_FAIL(c, HD_ERR_INVALID, "unknown debug buffer %s", name);
    const size_t n = it->second.second.first;
    if ((int64_t)n != n_elems) HD_FAIL(c, HD_ERR_INVALID, "debug write of %s needs %zu elements", name, n);
    HIPCHECK(c, hipSetDevice(c->device));
    HIPCHECK(c, hipDeviceSynchronize());
    if (it->second.second.second) {                        // bf16 buffer: round to nearest even, as the kernels do
        std::vector<unsigned short> tmp(n);
        for (size_t i = 0; i < n; ++i) {
            unsigned u; memcpy(&u, &host_in[i], 4);
            if ((u & 0x7fffffffu) > 0x7f800000u) { tmp[i] = (unsigned short)((u >> 16) | 0x40u); continue; }   // NaN stays NaN
            tmp[i] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
        }
        HIPCHECK(c, hipMemcpy(it->second.first, tmp.data(), n * 2, hipMemcpyHostToDevice));
    } else {
        HIPCHECK(c, hipMemcpy(it->second.first, host_in, n * sizeof(float), hipMemcpyHostToDevice));
    }
    return HD_OK;
}

int hd_set_option(hd_ctx* c, const char* key, int value) {
    if (!c || !key) return HD_ERR_INVALID;
    const std::string k = key;
    if (k == "xcd") c->xcd_on = value != 0;
    else if (k == "xcd2") c->xcd2_on = value != 0;
    else if (k == "xcd_phase_limit") c->xcd_phase_limit = value;
    else if (k == "xcd_force_global") c->xcd_force_global = value;
    else if (k == "face") c->face_on = value != 0;
    else if (k == "face_block_limit") c->face_block_limit = value;
    else if (k == "stage_limit_first") c->stage_limit_first = value;
    else if (k == "stage_test_abort") c->stage_test_abort = value;   // fault injection: 1..: XCD stages, group 0 gives up its wait for phase value - 1; 1000 + b: face stages, face 0, block b
    else HD_FAIL(c, HD_ERR_INVALID, "unknown option %s", key);
    c->graphs_valid = false;                               // captured graphs hold the old choice
    for (auto& kv : c->ws_cache) kv.second.graphs_valid = false;
    return HD_OK;
}
int hd_get_option(hd_ctx* c, const char* key) {
    if (!c || !key) return HD_ERR_INVALID;
    const std::string k = key;
    if (k == "xcd") return (c->xcd_ok && c->xcd_on) ? 1 : 0;
    if (k == "xcd2") return (c->xcd_ok && c->xcd_on && c->xcd2_on) ? c->xcd2_mask : 0;
    if (k == "xcd_stages") return (int)c->xstages.size();
    if (k == "face_stages") return (int)c->fstages.size();
    return HD_ERR_INVALID;
}

// To be called after the caller has synchronised the stream its hd_eps / hd_sample calls ran on: reports (once) a persistent
// stage that gave up during one of them.  The reference raises RuntimeError synchronously (test_refiner.py:89-91);
// here the call is asynchronous, its result is NaN-poisoned on the device, and this is where the error surfaces on the host.
int hd_check(hd_ctx* c) {
    if (!c) return HD_ERR_INVALID;
    return check_xcd(c);
}

int hd_set_profiling(hd_ctx* c, int on) { if (!c) return HD_ERR_INVALID; c->profiling = on != 0; return HD_OK; }

int hd_get_profile(hd_ctx* c, double* loop_ms, double* step_ms_avg, int64_t* weight_bytes_per_step, double* flops_per_face_step) {
    if (!c) return HD_ERR_INVALID;
    float ms = 0.f;
    if (c->profiling && c->last_steps > 0) {
        HIPCHECK(c, hipEventSynchronize(c->ev1));
        HIPCHECK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    }
    if (loop_ms) *loop_ms = ms;
    if (step_ms_avg) *step_ms_avg = c->last_steps > 0 ? ms / c->last_steps : 0.0;
    if (weight_bytes_per_step) *weight_bytes_per_step = c->weight_bytes_per_step;
    if (flops_per_face_step) *flops_per_face_step = c->flops_per_face_step;
    return HD_OK;
}

}  // extern "C"
