"""Parameter manifest of the refiner sampling path.

This is the state-dict surface the drop-in must accept: the
same key names and shapes as the reference `FacialRefiner(latent_res)`
(/root/reference/models/refiner.py:10-16), written out from the architecture
description rather than from any module tree so that the C-ABI packer, the
oracle and the synthetic-weight generator all share one source of truth.

kinds (used by the synthetic initialiser and the packer):
  conv_w / lin_w : dense weight, fan_in = prod(shape[1:])
  dw_w           : depthwise 3x3 weight (2C,1,3,3), fan_in 9
  bias           : bias attached to a weight with the given fan_in
  ln_w / ln_b    : LayerNorm2d affine
  res_scale      : NAF beta / gamma (1,C,1,1)
  bn_w / bn_b / bn_mean / bn_var / bn_count : BatchNorm2d (eval mode)
"""
from collections import OrderedDict

WIDTH = 128
ENC_BLOCKS = (2, 2, 4, 8)
MID_BLOCKS = 8
DEC_BLOCKS = (2, 2, 2, 2)
TIME_DIM = WIDTH * 4            # 512, models/denoiser/model.py:150
RESNET_LAYERS = (3, 4, 6, 3)    # models/idc/model.py:165


def _conv(out, name, cout, cin, kh, kw, bias=True):
    fan_in = cin * kh * kw
    out[name + ".weight"] = ((cout, cin, kh, kw), "conv_w", fan_in)
    if bias:
        out[name + ".bias"] = ((cout,), "bias", fan_in)


def _linear(out, name, cout, cin):
    out[name + ".weight"] = ((cout, cin), "lin_w", cin)
    out[name + ".bias"] = ((cout,), "bias", cin)


def _bn(out, name, c):
    out[name + ".weight"] = ((c,), "bn_w", 0)
    out[name + ".bias"] = ((c,), "bn_b", 0)
    out[name + ".running_mean"] = ((c,), "bn_mean", 0)
    out[name + ".running_var"] = ((c,), "bn_var", 0)
    out[name + ".num_batches_tracked"] = ((), "bn_count", 0)


def _naf_block(out, p, c, time_dim=None):
    # parameter registration order of ConditionalNAFBlock / NAFBlock
    # (models/denoiser/conditional_naf.py:14-101, models/fpg/naf.py:24-103)
    out[p + ".beta"] = ((1, c, 1, 1), "res_scale", 0)
    out[p + ".gamma"] = ((1, c, 1, 1), "res_scale", 0)
    if time_dim:
        _linear(out, p + ".mlp.1", 4 * c, time_dim // 2)
    _conv(out, p + ".conv1", 2 * c, c, 1, 1)
    out[p + ".conv2.weight"] = ((2 * c, 1, 3, 3), "dw_w", 9)
    out[p + ".conv2.bias"] = ((2 * c,), "bias", 9)
    _conv(out, p + ".conv3", c, c, 1, 1)
    _conv(out, p + ".sca.1", c, c, 1, 1)
    _conv(out, p + ".conv4", 2 * c, c, 1, 1)
    _conv(out, p + ".conv5", c, c, 1, 1)
    for n in ("norm1", "norm2"):
        out[p + "." + n + ".weight"] = ((c,), "ln_w", 0)
        out[p + "." + n + ".bias"] = ((c,), "ln_b", 0)


def _hca(out, p, c):
    # models/fpg/hca.py:6-23
    _linear(out, p + ".channel_mlp.0", c, c)
    _linear(out, p + ".channel_mlp.2", c, c)
    _conv(out, p + ".spatial_mlp.0", c // 2, c, 1, 1)
    _bn(out, p + ".spatial_mlp.1", c // 2)
    _conv(out, p + ".spatial_mlp.3", 1, c // 2, 1, 1)
    _bn(out, p + ".spatial_mlp.4", 1)
    _conv(out, p + ".fused_mlp.0", c, c, 3, 3)
    _bn(out, p + ".fused_mlp.1", c)


def denoiser_manifest(latent_res, prefix="denoiser", fused=True):
    """FusedDenoiser / Denoiser (models/denoiser/model.py:137-215, 32-110)."""
    o = OrderedDict()
    p = prefix
    _linear(o, p + ".time_mlp.1", TIME_DIM * 2, WIDTH)
    _linear(o, p + ".time_mlp.3", TIME_DIM, TIME_DIM)
    _conv(o, p + ".intro", WIDTH, 4, 3, 3)
    _conv(o, p + ".ending", 4, WIDTH, 3, 3)
    c = WIDTH
    for i, n in enumerate(ENC_BLOCKS):
        for j in range(n):
            _naf_block(o, f"{p}.encoders.{i}.{j}", c, TIME_DIM)
        c *= 2
    # registration order of the reference: encoders, decoders, middle_blks, ups, downs, hcas,
    # idc_conv (the empty ModuleLists are created first, models/denoiser/model.py:178-200)
    c = WIDTH * 16
    for i, n in enumerate(DEC_BLOCKS):
        c //= 2
        for j in range(n):
            _naf_block(o, f"{p}.decoders.{i}.{j}", c, TIME_DIM)
    c = WIDTH * 16
    for j in range(MID_BLOCKS):
        _naf_block(o, f"{p}.middle_blks.{j}", c, TIME_DIM)
    for i in range(4):
        _conv(o, f"{p}.ups.{i}.0", c * 2, c, 1, 1, bias=False)
        c //= 2
    c = WIDTH
    for i in range(4):
        _conv(o, f"{p}.downs.{i}", 2 * c, c, 2, 2)
        c *= 2
    if fused:
        c = WIDTH * 16
        for i in range(5):
            _hca(o, f"{p}.hcas.{i}", c)
            c //= 2
        s = latent_res // 16
        _conv(o, p + ".idc_conv", 2048 * s * s, 2048, 1, 1)
    return o


def fpg_manifest(prefix="fpg"):
    """FacialPriorGuidance (models/fpg/model.py:7-44)."""
    o = OrderedDict()
    p = prefix
    _conv(o, p + ".intro", WIDTH, 4, 3, 3)
    c = WIDTH
    for i, n in enumerate(ENC_BLOCKS):
        for j in range(n):
            _naf_block(o, f"{p}.encoders.{i}.{j}", c, None)
        c *= 2
    c = WIDTH
    for i in range(4):
        _conv(o, f"{p}.downs.{i}", 2 * c, c, 2, 2)
        c *= 2
    _conv(o, p + ".convs.0.0", c, c, 1, 1, bias=False)
    for i in range(1, 5):
        _conv(o, f"{p}.convs.{i}.0", c * 2, c, 1, 1, bias=False)
        c //= 2
    return o


def idc_manifest(prefix="idc"):
    """ResNet-50 without fc (models/idc/model.py:102-166); convs carry a bias AND a BN."""
    o = OrderedDict()
    p = prefix
    _conv(o, p + ".conv1", 64, 3, 7, 7, bias=False)
    _bn(o, p + ".batch_norm1", 64)
    cin = 64
    for li, (nblk, planes) in enumerate(zip(RESNET_LAYERS, (64, 128, 256, 512)), start=1):
        for b in range(nblk):
            q = f"{p}.layer{li}.{b}"
            _conv(o, q + ".conv1", planes, cin, 1, 1)
            _bn(o, q + ".batch_norm1", planes)
            _conv(o, q + ".conv2", planes, planes, 3, 3)
            _bn(o, q + ".batch_norm2", planes)
            _conv(o, q + ".conv3", planes * 4, planes, 1, 1)
            _bn(o, q + ".batch_norm3", planes * 4)
            if b == 0:
                _conv(o, q + ".i_downsample.0", planes * 4, cin, 1, 1)
                _bn(o, q + ".i_downsample.1", planes * 4)
            cin = planes * 4
    return o


def refiner_manifest(latent_res=16):
    """All keys of FacialRefiner(latent_res).state_dict() (models/refiner.py:14-16)."""
    o = OrderedDict()
    o.update(idc_manifest("idc"))
    o.update(denoiser_manifest(latent_res, "denoiser", fused=True))
    o.update(fpg_manifest("fpg"))
    return o


CR_WIDTH = 32                       # models/cr/model.py:38
CR_ENC = (2, 2, 4, 8)               # NAF blocks per encoder stage (models/cr/model.py:61-66)
CR_MID = 8
CR_DEC = (2, 2, 2, 2)


def _stn(out, p, c, res):
    """STNBlock(in_ch, in_res) (models/cr/stn.py:9-41): two valid convs with 2x2 max-pools, two Linears."""
    k0, k1 = (3, 1) if res <= 8 else (5, 3) if res <= 16 else (7, 5) if res <= 32 else (9, 7)
    fc_res = (res - k0 - 2 * k1 + 3) // 4
    fc = 10 * fc_res * fc_res
    _conv(out, p + ".localization.0", 8, c, k0, k0)
    _conv(out, p + ".localization.3", 10, 8, k1, k1)
    _linear(out, p + ".fc_loc.0", int(fc ** 0.5), fc)
    _linear(out, p + ".fc_loc.2", 6, int(fc ** 0.5))


def cr_stages():
    """(name, channels, resolution, naf blocks, sampling) of the nine NAF_STN_Blocks in execution order."""
    st = []
    c, r = CR_WIDTH, 128
    for i, n in enumerate(CR_ENC):
        st.append((f"encoders.{i}", c, r, n, "down"))
        c, r = c * 2, r // 2
    st.append(("middle_blocks", c, r, CR_MID, None))
    for i, n in enumerate(CR_DEC):
        st.append((f"decoders.{i}", c, r, n, "up"))
        c, r = c // 2, r * 2
    return st


def cr_manifest(prefix=""):
    """CoarseRestoration().state_dict() (models/cr/model.py:33-71), registration order."""
    o = OrderedDict()
    q = prefix + "." if prefix else ""
    _conv(o, q + "intro", CR_WIDTH, 3, 3, 3)
    _conv(o, q + "outro", 3, CR_WIDTH, 3, 3)
    for name, c, r, n, samp in cr_stages():
        for j in range(n):
            _naf_block(o, f"{q}{name}.nfbs.{j}", c, None)
        _stn(o, f"{q}{name}.stn", c, r)
        if samp == "down":
            _conv(o, f"{q}{name}.sampling", 2 * c, c, 2, 2)
        elif samp == "up":
            _conv(o, f"{q}{name}.sampling.0", 2 * c, c, 1, 1, bias=False)
    return o


def naf_levels(latent_res):
    """(channels, side) of the five UNet levels: C_l = 128*2^l, side = L/2^l."""
    return [(WIDTH << l, latent_res >> l) for l in range(5)]


# ---------------------------------------------------------------------------------------------------------------
# AutoencoderKL of "stable-diffusion-2-1-base" (the `vae` of test_refiner.py:176-178).  diffusers 0.32.2
# state-dict surface: block_out_channels (128, 256, 512, 512), layers_per_block 2, latent_channels 4, norm_num_groups 32,
# one single-head attention (to_q / to_k / to_v / to_out.0 Linear) in each mid block.
VAE_CHANNELS = (128, 256, 512, 512)
VAE_SCALING = 0.18215


def _gn(out, name, c):
    out[name + ".weight"] = ((c,), "ln_w", 0)
    out[name + ".bias"] = ((c,), "ln_b", 0)


def _vae_resnet(out, p, cin, cout):
    _gn(out, p + ".norm1", cin)
    _conv(out, p + ".conv1", cout, cin, 3, 3)
    _gn(out, p + ".norm2", cout)
    _conv(out, p + ".conv2", cout, cout, 3, 3)
    if cin != cout:
        _conv(out, p + ".conv_shortcut", cout, cin, 1, 1)


def _vae_attn(out, p, c):
    _gn(out, p + ".group_norm", c)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        _linear(out, p + "." + n, c, c)


def vae_manifest():
    o = OrderedDict()
    _conv(o, "encoder.conv_in", 128, 3, 3, 3)
    cin = 128
    for i, c in enumerate(VAE_CHANNELS):
        for j in range(2):
            _vae_resnet(o, f"encoder.down_blocks.{i}.resnets.{j}", cin, c)
            cin = c
        if i < 3:
            _conv(o, f"encoder.down_blocks.{i}.downsamplers.0.conv", c, c, 3, 3)
    _vae_resnet(o, "encoder.mid_block.resnets.0", 512, 512)
    _vae_attn(o, "encoder.mid_block.attentions.0", 512)
    _vae_resnet(o, "encoder.mid_block.resnets.1", 512, 512)
    _gn(o, "encoder.conv_norm_out", 512)
    _conv(o, "encoder.conv_out", 8, 512, 3, 3)
    _conv(o, "quant_conv", 8, 8, 1, 1)
    _conv(o, "post_quant_conv", 4, 4, 1, 1)
    _conv(o, "decoder.conv_in", 512, 4, 3, 3)
    _vae_resnet(o, "decoder.mid_block.resnets.0", 512, 512)
    _vae_attn(o, "decoder.mid_block.attentions.0", 512)
    _vae_resnet(o, "decoder.mid_block.resnets.1", 512, 512)
    cin = 512
    for i, c in enumerate(VAE_CHANNELS[::-1]):
        for j in range(3):
            _vae_resnet(o, f"decoder.up_blocks.{i}.resnets.{j}", cin, c)
            cin = c
        if i < 3:
            _conv(o, f"decoder.up_blocks.{i}.upsamplers.0.conv", c, c, 3, 3)
    _gn(o, "decoder.conv_norm_out", 128)
    _conv(o, "decoder.conv_out", 3, 128, 3, 3)
    return o
