// hd_aux.hip -- CoarseRestoration (models/cr/model.py:33-88, models/cr/stn.py:9-52) and the VAE boundary (§8 f2:
// diffusers AutoencoderKL, test_refiner.py:78-83,93) on the refiner path's kernels: weight ingest, workspaces, launch programs
// and their C-ABI entry points.  Internals: hd_internal.hpp.
#include "hd_internal.hpp"

// =============================================================================== CoarseRestoration (§8 f1)
// models/cr/model.py:73-88: intro -> 4 x [NAF blocks, STN, down] (stage outputs are the skips) -> [8 NAF, STN]
// -> 4 x [(+ skip), NAF blocks, STN, up] -> outro.  Stage s works on level buffers of its own geometry
// (C = 32 << level, side 128 >> level); NAF blocks, down- and up-convs are the refiner path's launches.
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

int finalize_cr(hd_ctx* c) {
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

int finalize_vae(hd_ctx* c) {
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

extern "C" {

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

}  // extern "C"
