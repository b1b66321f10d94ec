// hd_internal.hpp -- the library's internals shared by its host-side translation units (hd_lib.hip: refiner context, workspaces,
// launch programs, sampler, C-ABI; hd_aux.hip: CoarseRestoration and the VAE boundary): context and weight types, allocation,
// strict manifest, packing, the GEMM / NAF-block / conv launch builders.  Types live in namespace hdi (one definition for every
// unit); the helper functions are internal to each unit that includes this file.
#pragma once
// hd_internal.hpp — host side of libhifidiff_hip.so: context, weight ingest (fold + pack), the launch
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
#include "hd_end.hpp"
#include "hd_vae.hpp"
#include "hd_stage_api.hpp"

using namespace hd;

namespace hdi {

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

}  // namespace hdi
using namespace hdi;

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
    bool end_fold = getenv("HD_NO_END_FOLD") == nullptr;   // the last HCA conv + the ending conv as one launch (hd_end.hpp); end_fused: cleared when its launch is refused
    bool end_fused = true;
    bool up_fold = true;                      // the last up conv as the entry of the level-0 decoder stage (HD_NO_UP_FOLD=1: its own launch)
    bool down_fold = true;                    // the down conv of level 0 as the entry of the level-1 encoder stage (HD_NO_DOWN_FOLD=1: its own launch)
    bool intro_fold = true;                   // the intro conv as the entry of the level-0 encoder stage (HD_NO_INTRO_FOLD=1 at context creation: its own launch)
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

// mode_hint: the caller's measured choice for its own shapes (the transitions below), unless HD_GEMM_MODE forces one
void add_gemm(hd_ctx* c, std::vector<Op>& prog, const std::string& name, GemmP p, LdKind lk, EpKind ek, int mode_hint = -1) {
    int t128 = choose_mode(p, ek == EK_GATE || ek == EK_DWGATE);   // (kernel mode; name kept for the capture list)
    if (mode_hint >= 0 && hd_env("HD_GEMM_MODE") == nullptr) t128 = mode_hint;
    {   // many rows, long K, LayerNorm -> gate or bf16 -> residual (latent 32, levels 2 / 3): the deep-prefetch tall kernel; the
        // launch falls back to the mode above when the run-time shape does not fit (per-face timesteps)
        static const bool no_deep = hd_env("HD_NO_DEEP") != nullptr;
        const bool kinds = (lk == LK_LN && ek == EK_GATE) || (lk == LK_BF16 && ek == EK_RESID);
        if (kinds && !no_deep && mode_hint < 0 && (ek == EK_GATE ? deep_shape_ok<true>(p) : deep_shape_ok<false>(p))) t128 |= 16;
    }
    {   // level 3 of latent 32 (1024 rows, K = 1024): the role-split wide kernel (hd_wide.hpp) for the LayerNorm -> gate and bf16 -> residual GEMMs;
        // the launch falls back to the mode above when the run-time shape does not fit (per-face timesteps)
        static const bool no_wide = hd_env("HD_NO_WIDE") != nullptr;
        const bool pair = (lk == LK_LN && ek == EK_GATE), plain = (lk == LK_BF16 && ek == EK_RESID);
        if (!no_wide && ((pair && (wide_shape_ok<true>(p) == 1 || wide_shape_ok<true>(p) == 3)) || (plain && wide_shape_ok<false>(p) == 1))) t128 |= 32;
    }
    {   // tuning aid: HD_OP_MODE="downs.0=5,ups=2" overrides the mode of every launch whose name contains the key (first match)
        static const std::string over = hd_env("HD_OP_MODE") ? hd_env("HD_OP_MODE") : "";
        for (size_t at = 0; at < over.size();) {
            const size_t end = over.find(',', at) == std::string::npos ? over.size() : over.find(',', at);
            const size_t eq = over.find('=', at);
            if (eq != std::string::npos && eq < end && name.find(over.substr(at, eq - at)) != std::string::npos) { t128 = atoi(over.c_str() + eq + 1); break; }
            at = end + 1;
        }
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
    op.skinny_affine = p.xcd_tile_affine && (t128 & 48) == 0 && t128 != 0 && t128 != 1 && t128 != 4;     // bit 16 = deep kernel, bit 32 = wide kernel (launch_tile routes those elsewhere); base modes 0/1/4 are the tall kernel
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
    bool strip_pool = false;
    static const bool no_fuse = hd_env("HD_NO_DWFUSE") != nullptr, no_chain = hd_env("HD_NO_CHAIN") != nullptr;
    static const bool no_strip = hd_env("HD_NO_STRIP") != nullptr || no_fuse || no_chain;      // the chain kernel adds the strip sums up
    const bool by_strips = ((C == 128 && lv.H == 32) || (C == 256 && lv.H == 16)) && (*x_np) * (*x_cnt) == C && *x_np <= 16 && !no_strip;
    if (dwgate_ok(HW) && !no_fuse && !by_strips) {
        // LN1 + FiLM -> conv1 (+bias) -> depthwise 3x3 -> SimpleGate -> G, pooled mean: one launch
        GemmP p = base_gemm(bw.conv1, M);
        p.A = lv.Xb; p.lda = C; film_fields(p, 0);
        p.stats_in = lv.sx; p.stats_np = *x_np; p.stats_cnt = *x_cnt;
        p.out = lv.G; p.ldo = C; p.dw_w = bw.dw_wT; p.dw_b = bw.dw_b; p.pooled = lv.pooled; p.pooled16 = lv.pooled16; p.side = lv.H;
        add_gemm(c, prog, bw.name + ".conv2_gate_pool", p, LK_LN, EK_DWGATE);
    } else if (by_strips) {
        // 32 x 32 faces at C = 128, 16 x 16 at C = 256 (latent 32, levels 0 / 1): the same fusion by strips of 4 image rows, depthwise conv out of
        // the MFMA accumulators (hd_strip.hpp)
        StripP q{};
        q.faces = M / HW; q.side = lv.H; q.C = C;
        q.Xb = lv.Xb; q.stats_in = lv.sx; q.stats_np = *x_np; q.stats_cnt = *x_cnt;
        q.W1 = bw.conv1.w; q.b1 = bw.conv1.bias; q.dw_wT = bw.dw_wT; q.dw_b = bw.dw_b;
        q.film = static_film; q.film_gain_off = bw.film_off + C; q.film_bias_off = bw.film_off; q.film_face_stride = 0; q.face0 = c->ch->face0; q.ln_eps = 1e-6f;
        q.G = lv.G; q.pool_part = lv.T1;                          // T1 itself is never written on this path: its first faces x 8 x C floats hold the strip sums
        Chain* chp = c->ch;
        Op op;
        op.name = bw.name + ".conv2_gate_pool"; op.out = lv.G; op.out_elems = (size_t)M * C; op.out_bf16 = 1;
        op.run = [c, chp, q](hipStream_t s) mutable -> hipError_t {
            StripP r = q;
            if (r.film == nullptr) { r.film = c->film_from_cur ? chp->film_cur : c->film_table; r.film_face_stride = c->film_face_stride; }
            return run_strip_dwgate(r, s);
        };
        prog.push_back(op);
        strip_pool = true;                                        // the chain kernel below adds the strip sums up itself (no pool_finish launch)
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
    if ((C == 128 || C == 256) && HW % 32 == 0 && !no_fuse && !no_chain) {        // also behind the unfused depthwise path (latent 32, level 0)
        // levels 0/1: sca -> conv3 -> residual -> LN+FiLM -> conv4 -> gate -> conv5 -> residual in ONE launch (hd_chain.hpp)
        ChainP q{};
        q.M = M; q.hw = HW; q.face0 = c->ch->face0;
        q.G = lv.G; q.pooled = lv.pooled; q.X = lv.X;
        if (strip_pool) { q.pool_part = lv.T1; q.pool_nparts = lv.H / 4 /* 8 or 4 strips per face */; q.pool_scale = 1.0f / (float)HW; q.pooled_out = lv.pooled; }
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
    // per-shape choices measured at batch 64, latent 32 (profiles/r04_mode_sweep_L32_blocks.txt), where choose_mode's rules (tuned on latent 16) are off:
    // level 3 (1024 rows, C = 1024) conv3 / conv5 on 32-row tiles (12.9 -> 12.1 us each); level 2 (4096 rows, C = 512) conv3 / conv4 on the
    // 64-row tall tile (17.8 -> 16.5, 16.5 -> 15.6 us)
    const int hint35 = (M == 1024 && C == 1024) ? 3 : -1, hint3 = (M == 4096 && C == 512) ? 1 : hint35, hint4 = (M == 4096 && C == 512) ? 1 : -1;
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
            add_gemm(c, prog, bw.name + ".conv3", p, LK_BF16, EK_RESID, hint3);
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
            add_gemm(c, prog, bw.name + ".conv3", p, LK_BF16S, EK_RESID, hint3);
        }
    }
    {   // LN2 + FiLM -> conv4 -> SimpleGate -> G2 (bf16, reuses G)
        GemmP p = base_gemm(bw.conv4, M);
        p.A = lv.Yb; p.lda = C; film_fields(p, 1);
        p.stats_in = lv.sy; p.stats_np = C / 32; p.stats_cnt = 32;
        p.out = lv.G; p.ldo = C;
        add_gemm(c, prog, bw.name + ".conv4", p, LK_LN, EK_GATE, hint4);
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
        add_gemm(c, prog, bw.name + ".conv5", p, LK_BF16, EK_RESID, hint35);
    }
    *x_np = C / 32; *x_cnt = 32;
}

void add_down(hd_ctx* c, std::vector<Op>& prog, const std::string& name, const PackedW& w, const Level& src, const Level& dst) {
    GemmP p = base_gemm(w, dst.M);                      // Conv2d(C, 2C, 2, 2) as a patch-gather GEMM
    p.A = src.Xb; p.lda = src.C; p.Hin = src.H; p.Win = src.H; p.Cin = src.C; p.KH = 2; p.KW = 2; p.stride = 2; p.pad = 0;
    p.Hout = dst.H; p.Wout = dst.H; p.ntaps = 4;          // gathers the bf16 copy conv5 wrote (same rounding point as before)
    p.out = dst.X; p.ldo = dst.C; p.stats_out = dst.sx; p.out16 = dst.Xb;
    // many output rows (latent 32, levels 0 / 1 at batch 64): choose_mode's M-split workgroups were tuned on the blocks' GEMMs; the patch gather
    // is faster on 32-row tiles (profiles/r04_mode_sweep_L32.txt: 33.5 -> 24.3 and 25.1 -> 22.8 us)
    add_gemm(c, prog, name, p, LK_CONV_BF16, EK_BIASF32, dst.M >= 4096 ? 3 : -1);
}

// 1x1 conv (no bias) + PixelShuffle(r) + skip add, written in place over the skip buffer
void add_up(hd_ctx* c, std::vector<Op>& prog, const std::string& name, const PackedW& w, const void* in, bool in_bf16, int M_in,
            int H_in, int C_in, float* out, const float* skip, int r, unsigned short* out16 = nullptr, float2* stats = nullptr) {
    GemmP p = base_gemm(w, M_in);
    p.A = in; p.lda = C_in; p.Hin = H_in; p.Win = H_in; p.shuffle_r = r;
    p.out = out; p.ldo = w.N / (r * r); p.resid = skip; p.bias = nullptr;
    p.out16 = out16; p.stats_out = stats;
    // latent 32 at batch 64 (profiles/r04_mode_sweep_L32.txt): 32-row tiles from 4096 input rows on (48.9 -> 37.2, 36.4 -> 26.5 us);
    // 1024 rows x K >= 1024: the 64-row tall tile (30.0 -> 22.8 us)
    const int hint = M_in >= 4096 ? 3 : (M_in >= 1024 && M_in < 2048 && w.Kp >= 1024) ? 1 : -1;
    add_gemm(c, prog, name, p, in_bf16 ? LK_BF16 : LK_F32, EK_PIXSHUF, hint);
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


// conv2.weight [2C][1][3][3] -> tap-major device copy for the fused conv1 epilogue
int make_dw_layout(hd_ctx* c, BlockW& bw) {
    float* t = nullptr;
    int rc = dev_alloc(c, &t, (size_t)2 * bw.C * 9);
    if (rc) return rc;
    hipLaunchKernelGGL(dw_weight_layout_kernel, dim3((2 * bw.C * 9 + 255) / 256), dim3(256), 0, 0, bw.dw_w, t, 2 * bw.C);
    HIPCHECK(c, hipGetLastError());
    bw.dw_wT = t;
    return HD_OK;
}


}  // namespace

// hd_aux.hip
int finalize_cr(hd_ctx* c);
int finalize_vae(hd_ctx* c);
