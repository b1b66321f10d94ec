// hd_lib.hip -- libhifidiff_hip.so: the refiner's context (weights, per-batch workspaces), the denoiser / conditioning launch
// programs, the hipGraph-replayed sampler and the C-ABI of include/hifidiff_hip.h.  Shared internals: hd_internal.hpp;
// CoarseRestoration and the VAE boundary: hd_aux.hip; the persistent stage kernels: hd_stages.hip.
#include "hd_internal.hpp"

static std::string g_create_error;

namespace {

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
// diffusion steps per captured graph (kGraphSteps; HD_GRAPH_STEPS under HD_EXPERIMENTS=1 measures other lengths)
static int graph_steps() {
    static const int v = [] { const char* e = hd_env("HD_GRAPH_STEPS"); const int n = e ? atoi(e) : kGraphSteps; return n < 1 ? 1 : (n > 250 ? 250 : n); }();
    return v;
}
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
    // The intro conv: a launch of its own, or -- latent 16, batch <= 64, face-cluster stages available -- the ENTRY of the level-0 encoder stage
    // (hd_face.hpp: every workgroup computes x of its own and halo image rows from the latents; one launch less per step).  The launch stays as
    // the first step of that stage's per-GEMM form.
    std::function<hipError_t(hipStream_t)> intro_run;
    {
        const float *lat = c->ch->lat, *w = c->intro_wT, *b = ib->dev; float* out = c->ch->lv[0].X; float2* sx = c->ch->lv[0].sx;
        unsigned short* xb = c->ch->lv[0].Xb;
        const int M = c->ch->lv[0].M;
        intro_run = [=](hipStream_t s) -> hipError_t {
            if (M >= kLongRunRows) hipLaunchKernelGGL(intro_conv_kernel<16>, dim3((M / 16 + 3) / 4), dim3(256), 0, s, lat, w, b, out, xb, sx, B, L, chp->step_state, c->advance);
            else hipLaunchKernelGGL(intro_conv_kernel<kIntroPx>, dim3((M / kIntroPx + 3) / 4), dim3(256), 0, s, lat, w, b, out, xb, sx, B, L, chp->step_state, c->advance);
            return hipGetLastError();
        };
    }
    const bool fold_intro = c->xcd_ok && c->face_ok && c->intro_fold && B <= 64 && L == 16 && c->ch->lv[0].C == 128 && c->ch->lv[0].H == 16;
    if (!fold_intro) {
        prog.push_back({"intro", intro_run});
        prog.back().out = c->ch->lv[0].X; prog.back().out_elems = (size_t)c->ch->lv[0].M * 128;
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
                const hipError_t e = run_xcd2_stage(l3 ? 1024 : 512, r, s);
                if (e == hipSuccess) return e;
                (void)hipGetLastError();                      // refused launch (LDS / CU budget of this device or tenant): nothing ran -> the K-split form, then the per-GEMM launches
                c->xcd2_on = false;
            }
            if (c->xcd_ok && c->xcd_on && c->chains.size() == 1 && c->film_face_stride == 0) {
                XStageP r = sp;
                r.film = c->film_from_cur ? chp->film_cur : c->film_table;
                r.phase_limit = (c->stage_limit_first < 0 || c->stage_limit_first == first) ? c->xcd_phase_limit : 0;
                r.force_global = c->xcd_force_global; r.test_abort = c->stage_test_abort;
                const hipError_t e = run_xcd_stage(l3 ? 1024 : 512, r, s);
                if (e == hipSuccess) return e;
                (void)hipGetLastError();                      // as the face stages below: a refused launch is recoverable
                c->xcd_on = false;
            }
            for (auto& o : *sub) { const hipError_t e = o.run(s); if (e != hipSuccess) return e; }
            return hipSuccess;
        };
        prog.push_back(op);
    };
    // Levels 0 / 1 (latent 16, batch <= 64): a run of blocks as ONE launch with the rows of a face split over a cluster of
    // workgroups (hd_face.hpp); the per-block launches (fused conv1 + chain kernel) stay as the other form of the same op.
    auto add_face_stage = [&](int nblk, const Level& lv, const GateOut* gate, bool want_xb, bool with_intro = false, const Op* down_op = nullptr,
                              const Op* up_op = nullptr, const unsigned short* up_A = nullptr) {
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
        // the intro conv as this stage's entry (the program then has no intro launch: it is the first step of the per-GEMM form below)
        // (level 1: the same for the down conv of level 0)
        const std::function<hipError_t(hipStream_t)> intro_first = with_intro ? intro_run : down_op ? down_op->run : up_op ? up_op->run : std::function<hipError_t(hipStream_t)>();
        if (up_op) { fp.up_A = up_A; fp.up_W = c->den_up[3].w; }        // level 0's decoder stage: the last up conv (X holds the encoder skip)
        if (with_intro) { fp.intro_lat = chp->lat; fp.intro_wT = c->intro_wT; fp.intro_b = ib->dev; fp.intro_step = &chp->step_state->step; }
        if (down_op) { fp.down_A = c->ch->lv[0].Xb; fp.down_W = c->den_down[0].w; fp.down_b = c->den_down[0].bias; }
        Op op;
        op.name = c->den_blocks[first + nblk - 1].name + ".conv5"; op.out = lv.X; op.out_elems = (size_t)lv.M * lv.C; op.out_bf16 = 0;
        op.run = [c, chp, fp, sub, c128, first, intro_first](hipStream_t s) -> hipError_t {
            // (per-face timesteps and split batches run the per-GEMM form: one FiLM row and every workgroup resident are what the stage needs)
            if (c->xcd_ok && c->face_on && c->chains.size() == 1 && c->film_face_stride == 0) {
                FStageP r = fp;
                r.film = c->film_from_cur ? chp->film_cur : c->film_table;
                r.block_limit = (c->stage_limit_first < 0 || c->stage_limit_first == first) ? c->face_block_limit : 0;
                r.test_abort = c->stage_test_abort;
                r.intro_advance = c->advance;
                const hipError_t e = run_face_stage(c128 ? 128 : 256, c128 ? 32 : c->face_l1_rows, r, s);
                if (e == hipSuccess) return e;
                (void)hipGetLastError();                      // (the dynamic-LDS grant was refused: nothing was launched) -> the per-block launches
                c->face_on = false;
            }
            if (intro_first) { const hipError_t e = intro_first(s); if (e != hipSuccess) return e; }
            for (auto& o : *sub) { const hipError_t e = o.run(s); if (e != hipSuccess) return e; }
            return hipSuccess;
        };
        prog.push_back(op);
        np = lv.C / 32; cnt = 32;
    };
    // the down conv of level 0 as the entry of the level-1 stage (same conditions as that stage; HD_NO_DOWN_FOLD=1 keeps the launch)
    const bool fold_down0 = fold_intro && c->down_fold && c->ch->lv[1].C == 256 && c->ch->lv[1].H == 8 && c->den_down[0].K == 512 && c->den_down[0].N == 256;
    std::vector<Op> down0;
    for (int l = 0; l < 4; ++l) {
        if (l >= 2) add_stage(enc[l], c->ch->lv[l], nullptr);
        else add_face_stage(enc[l], c->ch->lv[l], nullptr, true, l == 0 && fold_intro, (l == 1 && fold_down0) ? &down0[0] : nullptr);
        if (stage_rc) return stage_rc;
        add_down(c, (l == 0 && fold_down0) ? down0 : prog, "downs." + std::to_string(l), c->den_down[l], c->ch->lv[l], c->ch->lv[l + 1]);
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
    // (with the other folds: the program of the persistent stages; the one-launch-per-GEMM programs keep every launch and its tap)
    const bool fuse_end = cond && fold_intro && c->end_fold && L == 16 && c->ch->lv[0].C == 128 && c->ch->lv[0].H == 16 && !c->hca[4].centre_only;
    std::vector<Op> hca4;
    for (int i = 0; i < 4; ++i) {
        const int l = 3 - i;
        const Level &hi = c->ch->lv[l + 1], &lo = c->ch->lv[l];
        // x = PixelShuffle(up(x)) + enc_skip (model.py:249-251); the epilogue also leaves the bf16 copy and the
        // LayerNorm partials of the new rows (one per 32 channels)
        // the last up conv as the entry of the level-0 decoder stage (same conditions as that stage; HD_NO_UP_FOLD=1 keeps the launch)
        const bool fold_up = i == 3 && fold_intro && c->up_fold && lo.C == 128 && lo.H == 16 && c->den_up[3].K == 256 && c->den_up[3].N == 512;
        std::vector<Op> up3;
        add_up(c, fold_up ? up3 : prog, "ups." + std::to_string(i), c->den_up[i], cond ? hi.Yb : hi.Xb, true, hi.M, hi.H, hi.C, lo.X, lo.X, 2, lo.Xb, lo.sx);
        np = lo.C / 32; cnt = 32;
        GateOut g; g.gate_c = c->ch->gate_c[i + 1]; g.gate_s = c->ch->gate_s[i + 1];
        if (l >= 2) {
            add_stage(2, lo, cond ? &g : nullptr);
            if (stage_rc) return stage_rc;
        } else {
            // unconditional: the up conv / ending read the blocks' own output
            add_face_stage(2, lo, cond ? &g : nullptr, !cond, false, nullptr, fold_up ? &up3[0] : nullptr, cond ? hi.Yb : hi.Xb);
            if (stage_rc) return stage_rc;
        }
        // the last HCA conv and the ending conv as ONE launch (hd_end.hpp; latent 16, conditional refiner; HD_NO_END_FOLD=1 keeps the two launches):
        // the HCA op is then the first step of the ending op's two-launch form
        if (cond) add_hca(c, (i == 3 && fuse_end) ? hca4 : prog, "hcas." + std::to_string(i + 1), c->hca[i + 1], lo.Xg, lo.Y, i < 3 ? lo.Yb : nullptr, lo.M, lo.H);
    }
    {
        const float *X = cond ? c->ch->lv[0].Y : c->ch->lv[0].X, *w = c->ending_wT, *b = eb->dev; float* eps = c->ch->eps;
        const int M = c->ch->lv[0].M;
        Chain* chp = c->ch;
        EndP ep{};
        std::function<hipError_t(hipStream_t)> hca4_run;
        if (fuse_end) {
            ep.B = B; ep.Xg = c->ch->lv[0].Xg; ep.W = c->hca[4].fused.w; ep.bias = c->hca[4].fused.bias; ep.ewT = w; ep.eb = b; ep.eps = eps;
            hca4_run = hca4[0].run;
        }
        prog.push_back({"ending", [=](hipStream_t s) -> hipError_t {
                            // sampling loop (film_from_cur): the launch also applies the scheduler update to this
                            // chain's latents and stages the next step's FiLM row
                            SchedArgs sa{};
                            if (fuse_end && c->end_fused) {
                                EndP q = ep;
                                if (c->film_from_cur) {
                                    const size_t per_face = (size_t)4 * L * L;
                                    q.sa.lat = chp->lat; q.sa.coef = c->coef_dev; q.sa.st = chp->step_state;
                                    q.sa.elem0 = (int)(chp->face0 * per_face); q.sa.n_total = (int)(c->B * per_face);
                                    q.sa.film_table = c->film_table; q.sa.film_cur = chp->film_cur; q.sa.film_total = c->film_total;
                                }
                                const hipError_t e = launch_hca_ending(q, s);
                                if (e == hipSuccess) return e;
                                (void)hipGetLastError();              // the dynamic-LDS grant was refused: nothing was launched -> the two launches
                                c->end_fused = false;
                            }
                            if (hca4_run) { const hipError_t e = hca4_run(s); if (e != hipSuccess) return e; }
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
    c->intro_fold = getenv("HD_NO_INTRO_FOLD") == nullptr;
    c->down_fold = getenv("HD_NO_DOWN_FOLD") == nullptr;
    c->up_fold = getenv("HD_NO_UP_FOLD") == nullptr;
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
                    for (int r = 0; r < (multi ? graph_steps() : 1) && e == hipSuccess; ++r)
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
        for (; i + graph_steps() <= n; i += graph_steps())
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
    else if (k == "stage_test_abort") c->stage_test_abort = value;   // fault injection: 1..: XCD stages, group 0 gives up its wait for phase value - 1; 1000 + b: face stages, face 0, block b; 2000 + p: a loader wave of hd_xcd2.hpp, phase p
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
    if (k == "intro_fold") return (c->xcd_ok && c->face_ok && c->intro_fold) ? 1 : 0;
    if (k == "end_fold") return (c->end_fold && c->end_fused) ? 1 : 0;
    if (k == "up_fold") return (c->xcd_ok && c->face_ok && c->intro_fold && c->up_fold) ? 1 : 0;
    if (k == "down_fold") return (c->xcd_ok && c->face_ok && c->intro_fold && c->down_fold) ? 1 : 0;
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
