/*
 * hifidiff_hip.h — C-ABI of the MI355X-native HifiDiff refiner sampling path.
 *
 * The reference (js43o/HifiDiff) has no FFI/plugin interface: its boundary for this path is the
 * Python nn.Module surface called from the sampling loop.  Each entry point below names the
 * reference interface it replaces (file:line into the reference tree).  Plain pointers and sizes
 * only; no torch types.  All tensors are fp32, NCHW-contiguous, DEVICE pointers unless stated.
 * Every call enqueues its work on `stream` (a hipStream_t, passed as void*) and returns without
 * synchronising, except where noted.  Return value: 0 on success, negative hd_status on error;
 * the message is available from hd_last_error().  A context is not thread-safe; use one per device.
 */
#ifndef HIFIDIFF_HIP_H
#define HIFIDIFF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hd_ctx hd_ctx;

enum hd_status {
    HD_OK = 0,
    HD_ERR_INVALID = -1,   /* bad argument / shape / state              */
    HD_ERR_HIP = -2,       /* a HIP runtime call failed                  */
    HD_ERR_WEIGHTS = -3,   /* missing / unexpected / mis-shaped tensor   */
    HD_ERR_NOT_READY = -4  /* weights not loaded or batch not prepared   */
};

/* One state-dict entry.  Replaces `model.load_state_dict(load_file(ckpt))`
 * (test_refiner.py:162-164, models/refiner.py:18-25): same key names and shapes as
 * FacialRefiner(latent_res).state_dict(). */
typedef struct hd_tensor_desc {
    const char* name;       /* e.g. "denoiser.middle_blks.3.conv1.weight" */
    const void* data;       /* fp32 (int64 for *.num_batches_tracked, ignored) */
    int32_t ndim;
    int64_t shape[4];
    int32_t is_device;      /* 0: host pointer, 1: device pointer on the context's device */
} hd_tensor_desc;

/* Reverse-diffusion schedule in coefficient form (one row per step, host memory).
 * Replaces `scheduler.set_timesteps(n)` + `scheduler.step(eps, t, x)` of diffusers 0.32.2 as used at
 * test_refiner.py:85-91 (DDIM, eta 0) and the DDPM step of BASELINE's 1000-step configuration:
 *     x0     = clamp((x - c[0]*eps) / c[1], -c[2], +c[2])
 *     x_prev = c[3]*x0 + c[4]*x + c[5]*eps + c[6]*z          z ~ N(0, I)
 */
typedef struct hd_schedule {
    int32_t n_steps;
    const float* timesteps;   /* [n_steps] value fed to the time embedding at each step          */
    const float* coef;        /* [n_steps][7]                                                     */
} hd_schedule;

/* FacialRefiner(latent_res) (models/refiner.py:11-16): builds the network description for latent
 * side `latent_res` (16 for 16->128 px, 32 for 32->256 px) on HIP device `device`. */
int hd_create(hd_ctx** out, int latent_res, int device);
/* The unconditional pre-training network `Denoiser(latent_size)` (models/denoiser/model.py:32-134; sampled by
 * pretrain_denoiser.py:76-120): the same UNet without priors, HCAs and identity term.  Its state-dict keys are
 * passed with the prefix "denoiser." (denoiser.time_mlp.1.weight, denoiser.encoders.0.0.conv1.weight ...).
 * Use hd_prepare_unconditional instead of hd_prepare; hd_eps / hd_sample work as for the refiner. */
int hd_create_unconditional(hd_ctx** out, int latent_res, int device);
/* The coarse-restoration network that produces `cr_face` before the loop:
 * `CoarseRestoration()` (models/cr/model.py:33-88; called as `cr_module(ln_face)` at test_refiner.py:77,
 * infer_cr.py:48-60).  A context of its own: load its state dict (keys as in the reference, no prefix) with
 * hd_load_weights / hd_finalize_weights, then
 *   hd_cr_forward(ctx, B, ln_face [B,3,128,128], cr_face_out [B,3,128,128], stream). */
int hd_cr_create(hd_ctx** out, int device);
int hd_cr_forward(hd_ctx* ctx, int batch, const float* ln_face, float* cr_face_out, void* stream);

/* VAE boundary either side of the loop: replaces
 *   cr_latent = vae.encode(F.interpolate(cr_face, image_res, mode="bicubic")).latent_dist.sample() * 0.18215
 *                                                              (test_refiner.py:78-83; train_refiner.py:72-83 with to_vae_range)
 *   images    = vae.decode(latent / 0.18215).sample            (test_refiner.py:93; train_refiner.py:122-123)
 * with diffusers' AutoencoderKL of "stable-diffusion-2-1-base" (third party; weights come from the caller's state dict with
 * diffusers' key names, loaded with hd_load_weights / hd_finalize_weights on a context from hd_vae_create).
 *   hd_vae_encode: images [B,3,in_res,in_res] fp32 NCHW; bicubic (align_corners=False) to image_res when they differ;
 *                  vae_range 1 applies clamp(0,1)*2-1 first; moments_out [B,8,L,L] (mean | logvar of the posterior) and / or
 *                  latents_out [B,4,L,L] = (mean + std * z) * 0.18215 with z = noise [B,4,L,L] or device Philox(seed); L = image_res/8.
 *   hd_vae_decode: latents [B,4,L,L] -> images_out [B,3,8L,8L]. */
int hd_vae_create(hd_ctx** out, int device);
int hd_vae_encode(hd_ctx* ctx, int batch, int in_res, int image_res, const float* images, int vae_range, const float* noise, uint64_t seed,
                  float* moments_out, float* latents_out, void* stream);
int hd_vae_decode(hd_ctx* ctx, int batch, int latent_res, const float* latents, float* images_out, void* stream);
void hd_destroy(hd_ctx* ctx);
const char* hd_last_error(const hd_ctx* ctx);   /* ctx may be NULL: creation errors */

/* load_state_dict (strict): may be called several times with partial lists; hd_finalize_weights
 * checks that every key of FacialRefiner.state_dict() arrived with the right shape, folds eval-mode
 * BatchNorm into the adjacent conv, and packs all GEMM weights to bf16 MFMA-fragment order.
 * Synchronous (weight ingest is not on the hot path). */
int hd_load_weights(hd_ctx* ctx, const hd_tensor_desc* tensors, int n);
int hd_finalize_weights(hd_ctx* ctx);

/* Once-per-batch conditioning, hoisted out of the loop: `self.fpg(cr_latent)`, `self.idc(cr_face)`
 * (models/refiner.py:33-34), the HCA gates w_c / w_s (models/fpg/hca.py:26-27,33-48) and
 * `idc_conv(identity_embedding)` (models/denoiser/model.py:245).
 *   cr_latent [B,4,L,L]; cr_face [B,3,128,128] or NULL; id_emb [B,2048] or NULL (exactly one of
 *   cr_face / id_emb must be given: id_emb is what FusedDenoiser.forward receives directly). */
int hd_prepare(hd_ctx* ctx, int batch, const float* cr_latent, const float* cr_face,
               const float* id_emb, void* stream);
/* Unconditional Denoiser: `model(latents, t)` has nothing to hoist; this sizes the workspace for `batch`
 * latents and builds the launch program (pretrain_denoiser.py:101-110). */
int hd_prepare_unconditional(hd_ctx* ctx, int batch, void* stream);

/* Same, but from already-computed priors: FusedDenoiser.forward(latents, timesteps, facial_priors,
 * identity_embedding) (models/denoiser/model.py:217).  priors[i] is NCHW
 * [B, 2048>>i, (L/16)<<i, (L/16)<<i]. */
int hd_prepare_from_priors(hd_ctx* ctx, int batch, const float* const priors[5],
                           const float* id_emb, void* stream);

/* The two conditioning extractors on their own, for callers that use `model.fpg(cr_latent)` /
 * `model.idc(cr_face)` directly (models/refiner.py:33-34; FacialPriorGuidance.forward
 * models/fpg/model.py:46-64, ResNet.forward models/idc/model.py:122-135).
 *   priors_out[i]: NCHW [B, 2048>>i, (L/16)<<i, (L/16)<<i];  id_emb_out: [B,2048] (== (B,2048,1,1)). */
int hd_fpg(hd_ctx* ctx, int batch, const float* cr_latent, float* const priors_out[5], void* stream);
int hd_idc(hd_ctx* ctx, int batch, const float* cr_face, float* id_emb_out, void* stream);

/* One denoiser evaluation: FusedDenoiser.forward (models/denoiser/model.py:217-266) on the prepared
 * batch.  x, eps_out [B,4,L,L]; timesteps: n_t == 1 (shared) or n_t == B values, device fp32. */
int hd_eps(hd_ctx* ctx, const float* x, const float* timesteps, int n_t, float* eps_out, void* stream);

/* The whole reverse-diffusion loop (test_refiner.py:87-91 / train_refiner.py:111-120) on the
 * prepared batch, in latent space, x updated in place.  The per-step kernel sequence is captured
 * once into a hipGraph and replayed n_steps times.
 *   noise: [n_steps][B,4,L,L] device fp32 (z for every step; rows whose c[6]==0 are not read), or
 *          NULL to draw z on the device from Philox4x32-10(seed; step, element). */
int hd_sample(hd_ctx* ctx, float* x_inout, const hd_schedule* sched, const float* noise,
              uint64_t seed, void* stream);

/* One scheduler update on its own: `scheduler.step(eps, t, x).prev_sample` (test_refiner.py:91) in
 * the coefficient form of hd_schedule (coef7 on the host); x updated in place.  noise/seed/step as in
 * hd_sample.  Needs no context. */
int hd_scheduler_step(float* x_inout, const float* eps, const float* coef7, const float* noise,
                      uint64_t seed, int step, int64_t n_elems, void* stream);

/* Introspection for tests and profiling (not on the hot path; reads synchronise the device).
 * `which` selects the launch program: 0 = one denoiser evaluation (hd_eps / one hd_sample step),
 * 1 = the most recent hd_prepare prologue. */
int hd_num_ops(hd_ctx* ctx, int which);                    /* kernel launches in the program (one chain) */
int hd_num_chains(hd_ctx* ctx);                            /* concurrently scheduled sub-batches       */
int hd_debug_limit_ops(hd_ctx* ctx, int which, int n_ops); /* run only the first n ops (<0: all)      */
const char* hd_debug_op_name(hd_ctx* ctx, int which, int i);
/* copy the output buffer of op i to the host as fp32; host_out NULL -> just return the element count */
int64_t hd_debug_read_op(hd_ctx* ctx, int which, int i, float* host_out, int64_t max_elems);
/* copy a named internal buffer (DESIGN.md "Buffers") to the host as fp32; returns the element count */
int64_t hd_debug_read(hd_ctx* ctx, const char* name, float* host_out, int64_t max_elems);
/* overwrite a named internal buffer from host fp32 (bf16 buffers: rounded to nearest even): tests feed a launch the
 * oracle's value of its input ("teacher forcing") so that its own error is not buried under inherited drift */
int hd_debug_write(hd_ctx* ctx, const char* name, const float* host_in, int64_t n_elems);
/* Run-time switches that select between equivalent launch programs of the same arithmetic (tests compare them bit for
 * bit; no reference interface corresponds): "xcd" 1/0 = levels 2 / 3 as XCD-local persistent launches (hd_xcd.hpp) or one
 * launch per GEMM; "face" 1/0 the same for levels 0 / 1 (hd_face.hpp); "xcd_phase_limit" n / "face_block_limit" n = stop
 * the persistent stages after n phases / blocks (0: all), "stage_limit_first" i = only the stage whose first block has index
 * i (-1: every stage) -- tests read a stage's residual stream block by block; "xcd_force_global" 1 = its
 * placement-independent hand-off form.  hd_get_option: "xcd" (effective), "xcd_stages" (stages built so far). */
int hd_set_option(hd_ctx* ctx, const char* key, int value);
int hd_get_option(hd_ctx* ctx, const char* key);
/* Error status of the asynchronous calls.  hd_eps / hd_sample only enqueue work; a persistent stage launch that has to give
 * up a hand-off wait (a workgroup of the stage was not resident: another tenant on the GPU) fills THAT call's result
 * (eps_out / x_inout) with NaN on the device and raises a host-visible word.  hd_check() -- to be called after the caller
 * has synchronised the stream, e.g. where the reference's loop would have raised RuntimeError (test_refiner.py:89-91) --
 * returns HD_ERR_HIP once for such a call (hd_last_error() names the stage and phase) and switches the context to one
 * launch per GEMM; the next hd_eps / hd_sample performs the same check on entry.  0 when nothing failed.
 * Fault injection for tests: hd_set_option(ctx, "stage_test_abort", n): n in 1..: group 0 of every XCD-local stage gives up
 * its wait for phase n - 1; 1000 + b: face 0 of every face-cluster stage gives up the pool wait of block b; 2000 + p: the first
 * loader wave of group 0 of every autonomous-wave stage (hd_xcd2.hpp) gives up its wait before LayerNorm phase p (p = 0 or 3 mod 5);
 * 0: off.  Every call issued between a stage giving up and the check that reports it may be NaN-poisoned: the report names the
 * first failure only. */
int hd_check(hd_ctx* ctx);
/* HIP-event time in ms of the most recent hd_sample's replay loop (0 if profiling is off) and the
 * summed duration of the GEMM launches: used by bench.py for the roofline object */
int hd_set_profiling(hd_ctx* ctx, int on);
int hd_get_profile(hd_ctx* ctx, double* loop_ms, double* step_ms_avg, int64_t* weight_bytes_per_step,
                   double* flops_per_face_step);

#ifdef __cplusplus
}
#endif
#endif /* HIFIDIFF_HIP_H */
