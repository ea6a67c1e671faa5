/*
 * skgs.h -- C ABI of libskgs_hip.so: the MI355X (gfx950) implementation of SK_GS's per-frame hot path
 *           (LBS deform of every Gaussian + differentiable 3DGS tile rasterizer, forward and backward).
 *
 * Boundary: every entry point below replaces one pybind11 function of the reference's CUDA extension
 * `my_ext/_C` (resolved in Python through get_C_function(name), my_ext/_C/__init__.py:17-48), or -- for the
 * deform -- the torch/lietorch op sequence of networks/sk_gs.py.  Signatures carry only plain pointers, sizes
 * and a HIP stream: no torch types.  All tensors are fp32, contiguous, device memory unless stated; index
 * tensors are int32 (`radii`, `n_contrib`, top-k ids) or int64 (KNN indices, as pytorch3d returns them).
 *
 * Reference interface replaced (file:line relative to the reference root):
 *   skgs_rasterize_forward*      <- rasterize_gaussians               my_ext/_C/src/nerf/gaussian_rasterizer_forward.cu:260-317
 *   skgs_rasterize_backward      <- rasterize_gaussians_backward      my_ext/_C/src/nerf/gaussian_rasterizer_backwrad.cu:200-261,333
 *   skgs_rasterize_extra_forward <- gaussian_rasterize_extra_forward  my_ext/_C/src/nerf/gaussian_rasterizer_extra.cu:222-246
 *   skgs_rasterize_extra_backward<- gaussian_rasterize_extra_backward my_ext/_C/src/nerf/gaussian_rasterizer_extra.cu:248-277
 *   skgs_topk_weights            <- gaussian_topk_weights             my_ext/_C/src/nerf/gaussian_topk.cu:98-121
 *   skgs_mark_visible            <- mark_visible (commented out)      my_ext/_C/src/nerf/gaussian_rasterizer_imp.cu:75-103
 *   skgs_lbs_deform_forward/backward <- networks/sk_gs.py:1143-1150,1162,1192-1203 (+ lietorch SE3.act, lie.h:59-64,246)
 *   skgs_knn_bones               <- pytorch3d.ops.knn_points call     networks/sk_gs.py:757
 *   skgs_se3_blend_forward/backward <- (sk_T[indices].act(points[:, None]) * weights[..., None]).sum(1)  networks/sk_gs.py:1147,814
 *                                   as lietorch executes it (gather + act + mul + sum), tangent-space gradients (lie_cpu.cpp:217-236)
 *   skgs_lie_forward/backward    <- lietorch's backend ops (lie_expm ... lie_act4 and their backward)  my_ext/_C/src/ops_3d/lie_torch.cpp:357-385
 *   skgs_knn_dist_weights_*      <- calc_LBS_weight, `weighted_kernel` / `kernel` / `dist` branches  networks/sk_gs.py:757-766,770
 *
 * Opaque buffers (the reference's geomBuffer / binningBuffer / imgBuffer uint8 tensors, gaussian_render.h:118-158):
 * the caller allocates them with the sizes returned by skgs_*_buffer_bytes() and hands the same bytes back to the
 * backward / extra / top-k calls.  Their internal layout is private to this library (DESIGN.md "Data layout").
 *
 * Host synchronisation: the reference copies num_rendered to the host in the middle of the forward
 * (gaussian_rasterizer_forward.cu:209).  Here the forward is split so the caller decides:
 *   (a) skgs_rasterize_forward_stage1 -> sync -> read *host_num_rendered -> allocate binning buffer exactly ->
 *       skgs_rasterize_forward_stage2                     (reference behaviour, one sync)
 *   (b) skgs_rasterize_forward with a binning capacity the caller guessed; no sync; device-side overflow flag
 *       (skgs_read_status) tells later whether the guess was too small.       (hipGraph-capturable)
 *
 * Every function returns 0 on success, non-zero on error; skgs_last_error() returns the message of the last
 * failure on the calling thread (the reference raises through AT_ERROR -> Python RuntimeError).
 */
#ifndef SKGS_H_
#define SKGS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SKGS_VERSION 1
#define SKGS_TILE 16             /* BLOCK_X = BLOCK_Y, gaussian_render.h:29-30 */
#define SKGS_MAX_RENDER_EXTRA 4  /* E <= 4 in renderCUDA_{forward,backward}, gaussian_render.cu:117-139 */
/* Bone count up to which the one-launch skinning paths apply (skgs_knn_lbs_deform_forward, skgs_lbs_deform_backward_logits and
 * the LDS-staged dense logit gradient): they keep per-workgroup bone tables / gradient rows in LDS.  Beyond it (the 512
 * superpoints of the sp stage) callers use the separate entry points, which have no bone limit. */
#define SKGS_FUSED_LBS_MAX_BONES 60
#define SKGS_FUSED_LBS_MAX_K 8

typedef void* skgs_stream_t; /* hipStream_t */

/* The skeleton stage's deform as a job of the rasterizer's per-Gaussian launch (skgs_raster_inputs.deform_job): the arguments of
 * skgs_knn_lbs_deform_forward below (sk_gs.py:757-770 K nearest joints + softmax of the gathered sp_W logits, :1143-1150 linear
 * blend skinning, :1162,1192-1203 activations).  The Gaussian's mean / scale / rotation / opacity are computed in the lane that
 * projects them: written (with weights / indices) for the backward, never re-read by the forward.  Same arithmetic, same bits as
 * the separate launch.  Needs K <= min(8, M), M <= SKGS_FUSED_LBS_MAX_BONES, scales + rotations (no cov3D_precomp).
 * joints == NULL: the skinning ALONE (skgs_lbs_deform_forward's arithmetic) -- out_idx / out_weights are then INPUTS (the
 * superpoint stage's search has written them), sp_W is not read, any M, K <= 16, no row capacity. */
typedef struct skgs_knn_deform_job {
  int32_t M, K;
  const float* points;  /* [P,3] the positions the bones are searched from (xyz.detach(), sk_gs.py:1113) */
  const float* joints;  /* [M,3] */
  const float* sp_W;    /* [P,M] logits */
  const float* bone_T;  /* [M,7] */
  const float* bone_drot;   /* [M,4] */
  const float* bone_dscale; /* [M,3] */
  const float* xyz;         /* [P,3] */
  const float* log_scale;   /* [P,3] */
  const float* rot;         /* [P,4] */
  const float* opacity_logit; /* [P] */
  int64_t* out_idx;     /* [P,K] */
  float* out_weights;   /* [P,K] */
  float* means;         /* [P,3]  == skgs_raster_inputs.means3D   */
  float* scales;        /* [P,3]  == skgs_raster_inputs.scales    */
  float* rotations;     /* [P,4]  == skgs_raster_inputs.rotations */
  float* opacity;       /* [P]    == skgs_raster_inputs.opacity   */
  int32_t largest;      /* skinning alone (joints == NULL) only: see skgs_deform_inputs.largest */
} skgs_knn_deform_job;

/* Inputs of rasterize_gaussians / rasterize_gaussians_backward (same meaning, same order as the pybind args). */
typedef struct skgs_raster_inputs {
  int32_t P;          /* number of Gaussians */
  int32_t sh_degree;  /* active SH degree D (0..3) */
  int32_t sh_coeffs;  /* allocated SH coefficients per Gaussian M (sh is [P, M, 3]); 0 when colors_precomp is used */
  int32_t E;          /* extra channels blended with the colour (0..4) */
  int32_t image_height, image_width;
  float tanfovx, tanfovy, scale_modifier;
  int32_t prefiltered, debug, colmap;
  const float* viewmatrix; /* [4,4] colmap=1: transposed (column-major) world->view; colmap=0: row-major */
  const float* projmatrix; /* [4,4] same convention, full projection */
  const float* campos;     /* [3] */
  const float* means3D;    /* [P,3] */
  const float* opacity;    /* [P,1] */
  const float* sh;         /* [P,M,3] or NULL */
  const float* scales;     /* [P,3] or NULL */
  const float* rotations;  /* [P,4] xyzw, not normalised by the kernels (ops_3d.h:93-103) or NULL */
  const float* extras;     /* [P,E] or NULL */
  const float* colors_precomp; /* [P,3] or NULL */
  const float* cov3D_precomp;  /* [P,6] or NULL */
  /* ---- optional fused-epilogue inputs (NULL = the in-tree reference behaviour) ---- */
  const float* sh_rest;    /* split SH storage: `sh` is then the DC term [P,1,3] and sh_rest [P,M-1,3] (the reference's
                              _features_dc / _features_rest, concatenated by get_features, gaussian_splatting.py:170-173) */
  const float* background; /* [3]: out_color = C + T * bg inside the blend kernel (upstream diff_gaussian_rasterization
                              forward.cu renderCUDA epilogue; the in-tree variant composites in torch, sk_gs.py:1236) and
                              its dL/dT term in the backward */
  int32_t tile_bucket_capacity; /* 0: the reference's compact tile lists (count -> scan -> scatter).  Lcap > 0 ("bucket"
                              layout, skgs_rasterize_forward only): tile t owns the fixed slots [t Lcap, (t+1) Lcap) of a
                              binning buffer of T * Lcap instances, so the counting and scan launches disappear; a tile
                              with more than Lcap instances drops the excess and sets the overflow flag; num_rendered
                              and max_tile_count of skgs_status read -1. */
  const float* tanfov_device; /* NULL, or a DEVICE pointer to {tanfovx, tanfovy}: the kernels read the field of view there
                              instead of the two floats above.  With viewmatrix / projmatrix / campos (device pointers
                              already) this makes every camera parameter of a launch a device load, so ONE captured
                              hipGraph serves every training view: the caller rewrites a small "view slot" before each
                              replay (sk_gs_amd/view_slot.py; the reference builds its settings per view on the host,
                              networks/gaussian_splatting.py:271-284, train.py:179-250). */
  int32_t host_status_words;  /* stage 1: 0 / 1 -> host_num_rendered receives R (one int32, the reference's read-back); 3 ->
                              it receives {R, overflow flag, longest tile list} (three int32, the head of skgs_status) */
  int32_t longest_list_hint;  /* stage 2 / forward: 0 = unknown; > 0 = an upper bound of the longest tile list (what stage 1
                              just reported): the sort launches for longer lists, which would find nothing to do, are
                              skipped (~4.5 us each) */
  const int32_t* live_count;  /* NULL, or a DEVICE int32 n <= P: only Gaussians 0..n-1 exist.  P is then the CAPACITY the
                               * launches are sized for; rows n..P-1 get radius 0 and an all-zero record (what a culled
                               * Gaussian gets) and no gradient is written for them.  Densification (clone / split / prune,
                               * networks/gaussian_splatting.py:565-636) changes n in place: a captured hipGraph of the step
                               * stays valid -- nothing it baked in (pointers, grids, P) moved */
  const skgs_knn_deform_job* deform_job; /* NULL, or (forward only): means3D / scales / rotations / opacity are not read but
                               * COMPUTED by the per-Gaussian launch from this job and written there (see the struct) */
  int32_t tiles_per_gaussian_hint; /* 0 = unknown; > 0 = about how many tiles a Gaussian touches (R / P of a recent forward, with
                               * head room): the scatter launch chooses its lanes per Gaussian from it instead of from the CAPACITY
                               * of the tile lists (a bucket layout sized for one long list has many slots per Gaussian and few
                               * tiles per Gaussian: 8 lanes cost that launch 27 us where 4 take 16) */
} skgs_raster_inputs;

typedef struct skgs_raster_buffers {
  void* geom;    size_t geom_bytes;
  void* binning; size_t binning_bytes;
  void* img;     size_t img_bytes;
} skgs_raster_buffers;

/* status words kept in the geom buffer header, readable asynchronously */
typedef struct skgs_status {
  int32_t num_rendered;   /* R = sum of tiles touched (what the reference returns) */
  int32_t overflow;       /* 1 if R exceeded the binning capacity given to stage 2 */
  int32_t max_tile_count; /* longest per-tile list */
  int32_t overflow_events; /* sticky: incremented by every forward that overflowed; the library never resets it, so
                              it counts overflows since the caller last zeroed the first 16 bytes of the geom buffer */
} skgs_status;

size_t skgs_geom_buffer_bytes(int32_t P);
size_t skgs_img_buffer_bytes(int32_t image_width, int32_t image_height);
size_t skgs_binning_buffer_bytes(int64_t capacity /* tile instances */);
int64_t skgs_binning_capacity(size_t binning_bytes);

/* Forward, stage 1: per-Gaussian preprocess (cull, project, Sigma3D/2D, conic, radius, SH->RGB), per-tile counts,
 * exclusive scan.  Writes radii[P].  If host_num_rendered != NULL (pinned host memory) R is copied there
 * asynchronously on `stream`. */
int skgs_rasterize_forward_stage1(const skgs_raster_inputs* in, const skgs_raster_buffers* buf, int32_t* radii,
    int32_t* host_num_rendered, skgs_stream_t stream);
/* Forward, stage 2: scatter tile instances, per-tile depth sort, alpha-blend.  out_color [3,H,W], out_opacity [H,W]
 * (= 1 - T, no background: gaussian_render.cu:106-108), out_extra [E,H,W] or NULL. */
int skgs_rasterize_forward_stage2(const skgs_raster_inputs* in, const skgs_raster_buffers* buf, float* out_color,
    float* out_opacity, float* out_extra, skgs_stream_t stream);
/* Both stages back to back, no host synchronisation. */
int skgs_rasterize_forward(const skgs_raster_inputs* in, const skgs_raster_buffers* buf, int32_t* radii,
    float* out_color, float* out_opacity, float* out_extra, int32_t* host_num_rendered, skgs_stream_t stream);
/* Copies the status words to host memory (async on stream). */
int skgs_read_status(const skgs_raster_buffers* buf, skgs_status* host_status, skgs_stream_t stream);

typedef struct skgs_raster_grads {
  /* grad_outputs */
  const float* dL_dout_color;   /* [3,H,W] */
  const float* dL_dout_opacity; /* [H,W] or NULL (= 0) */
  const float* dL_dout_extra;   /* [E,H,W] or NULL */
  /* optional gradients chained in from gaussian_rasterize_extra_backward: added to the results (NULL = none) */
  const float* grad_means2D_in; /* [P,3] */
  const float* grad_conic_in;   /* [P,2,2] */
  const float* grad_opacity_in; /* [P,1] */
  /* outputs, all written completely (rows of culled Gaussians are zero).  With deform_backward_job / sp_skinning_job attached,
   * dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dscales and dL_drotations may each be NULL: the job takes those gradients over
   * in registers (80 bytes per Gaussian less to write) */
  float* dL_dmeans2D;   /* [P,3]  (NDC-scaled, z = 0) */
  float* dL_dconic;     /* [P,2,2] or NULL */
  float* dL_dcolors;    /* [P,3] */
  float* dL_dopacity;   /* [P,1] */
  float* dL_dmeans3D;   /* [P,3] */
  float* dL_dcov3D;     /* [P,6] */
  float* dL_dsh;        /* [P,M,3] or NULL when M == 0 */
  float* dL_dscales;    /* [P,3] */
  float* dL_drotations; /* [P,4] */
  float* dL_dextras;    /* [P,E] or NULL */
  float* dL_dsh_rest;   /* with skgs_raster_inputs::sh_rest: dL_dsh is [P,1,3] and this [P,M-1,3]; else NULL */
  /* scratch: P*16 floats, contents undefined on entry and exit ... */
  float* workspace; size_t workspace_bytes;
  /* ... unless workspace_is_zero != 0: the caller guarantees the scratch is all zero on entry and the library leaves
   * it all zero on exit (saves the clearing launch when the same scratch serves every step) */
  int32_t workspace_is_zero;
  /* Optional, instead of dL_dsh / dL_dsh_rest (which are then not written and may be NULL): [P,6] floats per Gaussian,
   * the unit view direction (x, y, z) and the clamp-masked colour gradient (r, g, b) -- the SH gradient of ONE view is
   * their outer product basis(dir) x g (gaussian_rasterizer_backwrad.cu:26-127).  View-parallel training all-gathers these
   * 24 bytes per Gaussian and view instead of all-reducing 192, and rebuilds the rows with skgs_sh_grad_from_factors. */
  float* dL_dsh_factors;
  /* densification statistics of this view, updated by the same launch that writes dL_dmeans2D (NULL: not updated) -- the
   * arithmetic of skgs_densify_stats (add_densification_stats, gaussian_splatting.py:503-513 + the max_radii2D update,
   * sk_gs.py:1990-1997) without its launch: for radii > 0: max_radii2D = max(., radii), accum += |dL_dmeans2D.xy| *
   * stat_grad_multiplier, denom += 1 */
  float* stat_xyz_gradient_accum; /* [P,1] */
  float* stat_denom;              /* [P,1] */
  float* stat_max_radii2D;        /* [P] */
  float stat_grad_multiplier;     /* 1 / (the scale the backward was seeded with); 0 is read as 1 */
  const struct skgs_deform_backward_job* deform_backward_job; /* NULL, or: see the struct (below skgs_deform_inputs) */
  const struct skgs_sp_skinning_job* sp_skinning_job;         /* NULL, or: see the struct (beside skgs_sp_skinning_backward);
                                                                 not together with deform_backward_job */
} skgs_raster_grads;
size_t skgs_backward_workspace_bytes(int32_t P);

int skgs_rasterize_backward(const skgs_raster_inputs* in, const skgs_raster_buffers* buf, const int32_t* radii,
    const float* out_opacity, const skgs_raster_grads* g, skgs_stream_t stream);

/* Blend arbitrary per-Gaussian features with the saved buffers. extra [P,E]; pixel_extra [H*W, E] (pixel-major,
 * the reference labels this tensor {W,H,E}). */
int skgs_rasterize_extra_forward(int32_t W, int32_t H, int32_t P, int32_t E, const float* extra,
    const skgs_raster_buffers* buf, float* pixel_extra, skgs_stream_t stream);
/* grad_means2D [P,3], grad_conic [P,2,2], grad_opacity [P,1] are accumulated INTO (caller zero-fills fresh ones);
 * dL_dextra [P,E] is overwritten. */
int skgs_rasterize_extra_backward(int32_t W, int32_t H, int32_t P, int32_t E, const float* extra,
    const float* out_opacity, const float* grad_pixel_extra, const skgs_raster_buffers* buf, float* grad_means2D,
    float* grad_conic, float* grad_opacity, float* dL_dextra, skgs_stream_t stream);
/* top_indices [H,W,k] int32 (-1 fill), top_weights [H,W,k] */
int skgs_topk_weights(int32_t topk, int32_t W, int32_t H, int32_t P, const skgs_raster_buffers* buf,
    int32_t* top_indices, float* top_weights, skgs_stream_t stream);
/* Parity-test hook (no reference counterpart): re-run the blend forward (renderCUDA_forward, gaussian_render.cu:16-112)
 * over the buffers of a finished forward and ALSO write, per pixel, a fingerprint of the list entries it blended --
 * census [H*W][2] = {number of entries blended, sum over them of mix(1-based list position)}, mix(k) =
 * (k * 2654435761) ^ (k >> 5) in uint32 -- so that a test can find exactly the pixels whose branch decisions (power > 0,
 * alpha < 1/255, T < 1e-4) differ from the CPU oracle's.  out_color [3,H,W] / out_opacity [H,W] are the same walk's
 * image (no background); n_contrib in the img buffer is rewritten with identical values. */
int skgs_render_census(int32_t W, int32_t H, const skgs_raster_buffers* buf, float* out_color, float* out_opacity,
    uint32_t* census, skgs_stream_t stream);
/* present [P] bytes (0/1): near-plane test of in_frustum{,_colmap} */
int skgs_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, int32_t colmap, uint8_t* present,
    skgs_stream_t stream);

/* dL/dsh rows from the factors of n_views views ([n_views][P][6], see skgs_raster_grads::dL_dsh_factors), views summed in
 * index order: dL_dsh [P,M,3], or the DC part [P,1,3] with dL_dsh_rest [P,M-1,3].  One view reproduces the rows
 * skgs_rasterize_backward writes, bit for bit. */
int skgs_sh_grad_from_factors(int32_t P, int32_t n_views, int32_t sh_degree, int32_t sh_coeffs, const float* factors,
    float* dL_dsh, float* dL_dsh_rest, skgs_stream_t stream);

/* ---- LBS deform + activation epilogue (networks/sk_gs.py:1143-1150,1162,1192-1203) ---- */
typedef struct skgs_deform_inputs {
  int32_t P, K, M;
  const float* points;        /* [P,3] detached copy of xyz (may alias xyz) */
  const float* weights;       /* [P,K] */
  const int64_t* indices;     /* [P,K] bone ids */
  const float* bone_T;        /* [M,7] tx,ty,tz,qx,qy,qz,qw */
  const float* bone_drot;     /* [M,4] */
  const float* bone_dscale;   /* [M,3] */
  const float* xyz;           /* [P,3] */
  const float* log_scale;     /* [P,3] */
  const float* rot;           /* [P,4] */
  const float* opacity_logit; /* [P,1] */
  const int32_t* live_count;  /* NULL, or a DEVICE int32 n <= P (see skgs_raster_inputs.live_count): the fused forward /
                               * backward (skgs_knn_lbs_deform_forward, skgs_lbs_deform_backward_logits) skip rows >= n */
  int32_t largest;            /* warp_method `largest` (exps/d_nerf_sp_gs.yaml:32, networks/sk_gs.py:811-816,849-850): the POSITION of a
                               * Gaussian follows the one bone with its largest weight (the first of equal ones, torch.argmax) --
                               * d_xyz = T[indices[n, argmax_k weights[n,k]]](p) - p -- while the rotation / scale offsets stay blended
                               * with the K weights; the weights then receive no gradient through the position.  Honoured by
                               * skgs_lbs_deform_forward, the skinning-alone deform_job of skgs_rasterize_forward, and
                               * skgs_sp_skinning_backward / skgs_raster_grads.sp_skinning_job; every other entry point refuses it */
} skgs_deform_inputs;
/* skgs_lbs_deform_backward_logits (below) as a job of skgs_rasterize_backward (skgs_raster_grads.deform_backward_job): the
 * per-Gaussian launch that produces dL/d(means3D, scales, rotations, opacity) hands them to the skinning backward in registers
 * (they are still written to the skgs_raster_grads outputs) -- the Gaussian's parameter gradients, its logit gradient and the
 * bone gradients come out of skgs_rasterize_backward, one launch less, the same arithmetic on the same values.  Needs
 * M <= 64, K <= 8, in->P / in->live_count equal to the rasterizer's, and means3D / scales / rotations / opacity of the
 * rasterizer being what skgs_lbs_deform_forward (or the deform_job of the forward) produced from `in`. */
typedef struct skgs_deform_backward_job {
  const skgs_deform_inputs* in;
  float* g_bone_T;        /* [M,7] */
  float* g_bone_drot;     /* [M,4] */
  float* g_bone_dscale;   /* [M,3] */
  float* g_xyz;           /* [P,3] */
  float* g_log_scale;     /* [P,3] */
  float* g_rot;           /* [P,4] */
  float* g_opacity_logit; /* [P,1] */
  float* g_sp_W;          /* [P,M] dense logit gradient, or NULL */
  float* g_logits;        /* [P,K] compact logit gradient, or NULL (one of the two is required) */
  void* workspace;        /* skgs_lbs_deform_backward_workspace_bytes(P, M) */
  size_t workspace_bytes;
} skgs_deform_backward_job;
/* means [P,3], scales [P,3], rotations [P,4] (normalised), opacity [P,1]; d_xyz/d_rot/d_scale optional (NULL) */
int skgs_lbs_deform_forward(const skgs_deform_inputs* in, float* means, float* scales, float* rotations,
    float* opacity, float* d_xyz, float* d_rot, float* d_scale, skgs_stream_t stream);
/* Every output is written completely (no zero-fill needed; the bone gradients are reduced without atomics, in a
 * fixed order, when M <= 64).  workspace: skgs_lbs_deform_backward_workspace_bytes(P, M) bytes of scratch. */
size_t skgs_lbs_deform_backward_workspace_bytes(int32_t P, int32_t M);
int skgs_lbs_deform_backward(const skgs_deform_inputs* in, const float* g_means, const float* g_scales,
    const float* g_rotations, const float* g_opacity, float* g_weights, float* g_bone_T, float* g_bone_drot,
    float* g_bone_dscale, float* g_xyz, float* g_log_scale, float* g_rot, float* g_opacity_logit, void* workspace,
    size_t workspace_bytes, skgs_stream_t stream);
/* skgs_lbs_deform_backward with the softmax backward of the `sp_W` weighting folded in (skgs_lbs_weights_backward /
 * _compact, networks/sk_gs.py:769-770): writes the dense logit gradient g_sp_W [P,M] and / or the compact one g_logits
 * [P,K] (either may be NULL, not both); g_weights may be NULL.  Needs M <= 64 and K <= 8.  Same values as the
 * two-call sequence. */
int skgs_lbs_deform_backward_logits(const skgs_deform_inputs* in, const float* g_means, const float* g_scales,
    const float* g_rotations, const float* g_opacity, float* g_weights, float* g_bone_T, float* g_bone_drot,
    float* g_bone_dscale, float* g_xyz, float* g_log_scale, float* g_rot, float* g_opacity_logit, float* g_sp_W,
    float* g_logits, void* workspace, size_t workspace_bytes, skgs_stream_t stream);
/* K (<= 16) nearest bones by squared L2 in `dim` dimensions, ascending, ties to the lower index. */
int skgs_knn_bones(int32_t P, int32_t M, int32_t K, int32_t dim, const float* points, const float* joints,
    float* out_dist, int64_t* out_idx, skgs_stream_t stream);
/* The two distance-based branches of calc_LBS_weight fused with the search (networks/sk_gs.py:757-766,770): K nearest bones in
 * `dim` (<= 16) dimensions as skgs_knn_bones, then
 *   kernel_radius != NULL:  w = (exp(-d / (2 r_i^2)) [* kernel_weight_i] + 1e-7) / sum_k(...)   (`kernel` / `weighted_kernel`;
 *                           r = exp(_sp_radius), kernel_weight = sigmoid(_sp_weight): the [M] activations stay with the caller)
 *   kernel_radius == NULL:  w = softmax_k(-d / temperature)                                    (`dist`)
 * out_idx [P,K] int64, out_weights [P,K], out_dist [P,K] (kept for the backward).
 * Backward: g_weights [P,K] -> g_points [P,dim] (optional), g_joints [M,dim], g_kernel_radius [M], g_kernel_weight [M]
 * (optional ones may be NULL): what autograd returns through knn_points' distances and the [indices] gathers.  The
 * workspace holds per-workgroup partial sums (skgs_knn_dist_weights_workspace_bytes).
 * raw_parameters != 0: kernel_radius / kernel_weight point at the RAW parameters `_sp_radius` / `_sp_weight` and the kernels
 * apply exp / sigmoid themselves (sk_gs.py:547-553), the backward returning the gradients w.r.t. the raw parameters: a step
 * without autograd then needs no launch for M values.  accumulate_joints != 0: g_joints += instead of = (the joints also get a
 * gradient through the kinematic chain). */
int skgs_knn_dist_weights_forward(int32_t P, int32_t M, int32_t K, int32_t dim, const float* points, const float* joints,
    const float* kernel_radius, const float* kernel_weight, float temperature, int32_t raw_parameters, int64_t* out_idx,
    float* out_weights, float* out_dist, skgs_stream_t stream);
size_t skgs_knn_dist_weights_workspace_bytes(int32_t P, int32_t M, int32_t dim);
int skgs_knn_dist_weights_backward(int32_t P, int32_t M, int32_t K, int32_t dim, const float* points, const float* joints,
    const float* kernel_radius, const float* kernel_weight, float temperature, int32_t raw_parameters,
    int32_t accumulate_joints, const float* weights, const int64_t* indices, const float* nn_dist, const float* g_weights,
    float* g_points, float* g_joints, float* g_kernel_radius, float* g_kernel_weight, void* workspace, size_t workspace_bytes,
    skgs_stream_t stream);
/* LBS weights from the per-Gaussian logits, the `sp_W` branch of calc_LBS_weight (networks/sk_gs.py:769-770):
 * weights[P,K] = softmax_k(sp_W[p, indices[p,k]]).  The backward writes the DENSE gradient g_sp_W[P,M] (zeros for the
 * bones outside the K nearest), i.e. what autograd's gather backward accumulates into a zero tensor. */
int skgs_lbs_weights_forward(int32_t P, int32_t M, int32_t K, const float* sp_W, const int64_t* indices, float* weights,
    skgs_stream_t stream);
/* skgs_lbs_weights_backward in two halves (view-parallel training all-reduces the compact half: the KNN indices are the
 * same on every rank): g_logits [P,K] = w * (g_w - sum_j w_j g_w_j), then its expansion into the dense g_sp_W [P,M]. */
int skgs_lbs_weights_backward_compact(int32_t P, int32_t K, const float* weights, const float* g_weights, float* g_logits,
    skgs_stream_t stream);
int skgs_lbs_logits_scatter(int32_t P, int32_t M, int32_t K, const int64_t* indices, const float* g_logits, float* g_sp_W,
    skgs_stream_t stream);
/* skgs_knn_bones (dim = 3) + skgs_lbs_weights_forward in one launch: out_idx [P,K] int64, out_weights [P,K]. */
int skgs_knn_lbs_weights(int32_t P, int32_t M, int32_t K, const float* points, const float* joints, const float* sp_W,
    int64_t* out_idx, float* out_weights, skgs_stream_t stream);
int skgs_lbs_weights_backward(int32_t P, int32_t M, int32_t K, const float* weights, const int64_t* indices,
    const float* g_weights, float* g_sp_W, skgs_stream_t stream);
/* skgs_knn_lbs_weights + skgs_lbs_deform_forward in one launch (stage `sk`: calc_LBS_weight sk_gs.py:757-770 followed by
 * the skinning and activation epilogue :1143-1150,1192-1203 on the same Gaussians).  out_idx / out_weights [P,K] are
 * written for the backward; means / scales / rotations / opacity as skgs_lbs_deform_forward.  Bit-identical to the two
 * separate calls. */
int skgs_knn_lbs_deform_forward(int32_t P, int32_t M, int32_t K, const float* points, const float* joints, const float* sp_W,
    const float* bone_T, const float* bone_drot, const float* bone_dscale, const float* xyz, const float* log_scale,
    const float* rot, const float* opacity_logit, int64_t* out_idx, float* out_weights, float* means, float* scales,
    float* rotations, float* opacity, const int32_t* live_count /* may be NULL */,
    skgs_stream_t stream);

/* ---- the reference's own skinning expression, for the lietorch stand-in (sk_gs_amd/lietorch.py) ----
 *   out[n] = sum_k weights[n,k] * SE3(T[indices[n,k]]).act(points[n])        networks/sk_gs.py:1147,814,1478  (weights NULL: 1;
 *                                                                             K = 1, no weights: :816,1481 `spT[p2sp].act(points)`)
 * which lietorch executes as gather [P,K,7] + act (lie.h:59-64,246; the SE3 constructor normalises q, lie.h:45-47) + mul + sum.
 * Backward in lietorch's convention (my_ext/_C/src/ops_3d/lie_cpu.cpp:217-236): g_T [M,7] holds the LEFT-TANGENT gradient
 * (tau, phi) = sum over the gathering rows of c [I | -hat(X p)], c = weights * g_out, in slots 0..5 and 0 in slot 6 -- what the
 * reference's chain of lietorch ops above `sk_T` expects; g_weights [P,K] = g_out . X p (NULL to skip, required NULL without
 * weights); g_points [P,3] = sum_k c R (NULL to skip: the reference detaches the points, sk_gs.py:833,1113).  Any M: tables of
 * up to 1024 bones are staged (and their gradient rows summed) in LDS, one partial per workgroup added in a fixed order by a
 * second launch (the order INSIDE a workgroup is the LDS atomics'); larger ones use global atomics. */
int skgs_se3_blend_forward(int32_t P, int32_t K, int32_t M, const float* T, const int64_t* indices, const float* points,
    const float* weights, float* out, skgs_stream_t stream);
size_t skgs_se3_blend_backward_workspace_bytes(int32_t P, int32_t M);
int skgs_se3_blend_backward(int32_t P, int32_t K, int32_t M, const float* T, const int64_t* indices, const float* points,
    const float* weights, const float* g_out, float* g_T, float* g_weights, float* g_points, void* workspace,
    size_t workspace_bytes, skgs_stream_t stream);

/* out [M,C] = sum over the R rows r of rows[r, :] filed under indices[r] (int64 in [0, M)): the backward of the reference's
 * `table[indices]` gathers of per-bone rows (networks/sk_gs.py:1148-1149 `sk_d_rot[indices]`, `sk_d_scale[indices]`, :760-763
 * `kernel_radius[indices]`, `kernel_weight[indices]`), R = P * K.  torch's index backward sorts the indices and walks every bone's
 * duplicates serially (7.5 ms per gather at 100k x 5 rows into 20 bones on an MI355X); here: LDS rows per workgroup, one partial
 * per workgroup, added in workgroup order by a second launch (tables beyond 48 KB: global atomics).  `out` is written completely. */
size_t skgs_index_add_rows_workspace_bytes(int64_t R, int32_t C, int32_t M);
int skgs_index_add_rows(int64_t R, int32_t C, int32_t M, const int64_t* indices, const float* rows, float* out, void* workspace,
    size_t workspace_bytes, skgs_stream_t stream);

/* ---- Lie-group operators for the lietorch stand-in (sk_gs_amd/lietorch.py): lietorch's backend, one launch per operator ----
 * What `lietorch_backends` is to upstream lietorch and my_ext/_C/src/ops_3d/lie_{cpu.cpp,gpu.cu,torch.cpp} (pybind `lie_expm`,
 * `lie_logm`, `lie_inv`, `lie_mul`, `lie_adj`, `lie_adjT`, `lie_act`, `lie_act4` + `_backward`, `lie_projector`;
 * lie_torch.cpp:357-385) restate in the reference tree.  group: lietorch's id, 1 = SO3 (K = 3 tangent, N = 4 embedding, q_xyzw),
 * 3 = SE3 (K = 6 = (tau, phi), N = 7 = (t, q)).  Rows are contiguous [B, width]; one lane per row.
 *   op  forward                         X      Y      out   | backward: grad (out's width) -> dX, dY (either may be NULL)
 *   0   exp   a -> X                    [K]    -      [N]   | da = dX[:K] J_l(a)                                  lie_cpu.cpp:25-38
 *   1   log   X -> a                    [N]    -      [K]   | dX = da J_l^-1(log X)                               :54-67
 *   2   inv                             [N]    -      [N]   | dX = -dY Adj(X^-1)                                  :84-97
 *   3   mul   X * Y                     [N]    [N]    [N]   | dX = dZ, dY = dZ Adj(X)                             :111-126
 *   4   adj   Adj(X) a                  [N]    [K]    [K]   | dX = -db adj(Adj(X) a), da = db Adj(X)              :143-162
 *   5   adjT  Adj(X)^T a                [N]    [K]    [K]   | dX = -a adj(Adj(X) db), da = Adj(X) db              :179-198
 *   6   act   X p                       [N]    [3]    [3]   | dX = dq [I | -hat(X p)], dp = dq R                  :217-236
 *   7   act4  X p (homogeneous)         [N]    [4]    [4]   | dX = dq act4_jacobian(X p), dp = dq Matrix4x4       :288-309
 *   8   vec()        (backward only: the forward is the identity)   dX = grad J,  J = orthogonal_projector(X)     lie.h:82-90,303-311
 *   9   InitFromVec  (backward only)                                dX = grad pinv(J) = (tau, 4 J_q (phi - t x tau))
 * Gradients of group elements are LEFT-TANGENT row vectors in the first K of their N slots, the remaining slots written 0.  Every
 * constructor normalises the quaternion (lie.h:45-47); small-angle series below 1e-6 (lie.h:23). */
int skgs_lie_forward(int32_t group, int32_t op, int64_t B, const float* X, const float* Y, float* out, skgs_stream_t stream);
int skgs_lie_backward(int32_t group, int32_t op, int64_t B, const float* grad, const float* X, const float* Y, float* dX, float* dY,
    skgs_stream_t stream);

/* ---- bone chain (scope row a-3): joint rotations -> global bone transforms, one launch per direction ----
 * Replaces kinematic() + skeleton_warp_SE3() (networks/sk_gs.py:1069-1107,193-206; lietorch SE3 product lie.h:242-246).
 *   sk_r_raw [M,4]  network output BEFORE "+[0,0,0,1], normalize" (sk_gs.py:1076)      joints [M,3]
 *   global_T [7] or NULL (identity)                                                     bone_T  [M,7] out
 * Skeleton topology as three int32 device arrays: parents[M] (direct parent; parents[root] = root), the bones sorted
 * by depth (level_nodes[M]) and level_start[num_levels+1] (level 0 = {root}).  chain_A [M,7] is written by the forward
 * (may be NULL for inference) and consumed by the backward.  g_joints / g_global_T may be NULL.
 * frame_index: NULL, or a DEVICE int32: global_T (and g_global_T) are then the base of a [frames, 7] table and the kernels
 * use row *frame_index (the frame is chosen without a host-side pointer: one captured graph for all frames). */
int skgs_bone_chain_forward(int32_t M, int32_t root, const int32_t* parents, const int32_t* level_nodes,
    const int32_t* level_start, int32_t num_levels, const float* sk_r_raw, const float* joints, const float* global_T,
    float* bone_T, float* chain_A, const int32_t* frame_index, skgs_stream_t stream);
int skgs_bone_chain_backward(int32_t M, int32_t root, const int32_t* parents, const int32_t* level_nodes,
    const int32_t* level_start, int32_t num_levels, const float* sk_r_raw, const float* joints, const float* global_T,
    const float* chain_A, const float* g_bone_T, float* g_sk_r_raw, float* g_joints, float* g_global_T,
    const int32_t* frame_index, skgs_stream_t stream);

/* ---- fused training-image loss (scope row (f)-1): lambda_l1 * mean|x-y| + lambda_ssim * (1 - mean SSIM(x,y)) ----
 * Replaces networks/losses/ssim.py:20-62 + image_loss.py:6-32 as composed at networks/sk_gs.py:1524-1529 (11x11
 * Gaussian window, sigma 1.5, zero padding).  pred, gt: [C,H,W].  loss3 (device, 3 floats) = {total, L1 mean, SSIM
 * mean}.  The workspace (skgs_image_loss_workspace_bytes) carries the SSIM derivative maps from forward to
 * backward.  grad_loss: device scalar dL/dloss, or NULL for 1.  gt_index: NULL, or a DEVICE int32: gt is then a stack
 * [views, C, H, W] and the kernels compare with image *gt_index (target chosen on the device). */
size_t skgs_image_loss_workspace_bytes(int32_t C, int32_t H, int32_t W);
int skgs_image_loss_forward(int32_t C, int32_t H, int32_t W, const float* pred, const float* gt, const int32_t* gt_index,
    float lambda_l1, float lambda_ssim, float* loss3 /* may be NULL: see the backward */, void* workspace,
    size_t workspace_bytes, skgs_stream_t stream);
/* loss3: NULL, or where this call puts {total, L1 mean, SSIM mean} of the forward that filled `workspace` (a forward
 * called with loss3 = NULL skips its one-workgroup summation launch; one workgroup of the backward does it instead). */
int skgs_image_loss_backward(int32_t C, int32_t H, int32_t W, const float* pred, const float* gt, const int32_t* gt_index,
    float lambda_l1, float lambda_ssim, const float* grad_loss, const void* workspace, size_t workspace_bytes,
    float* dL_dpred, float* loss3, skgs_stream_t stream);
/* The backward for a caller that holds the two terms as SEPARATE autograd outputs -- the reference's loss dict keeps
 * losses['rgb'] = w_image * L1 and losses['ssim'] = w_ssim * (1 - SSIM) apart (networks/sk_gs.py:1527-1529, losses/build.py:55-64),
 * so autograd hands back one cotangent per term:  dL_dpred = *grad_l1 * d(L1 mean)/dpred + *grad_ssim * d(1 - SSIM mean)/dpred.
 * grad_l1 / grad_ssim: DEVICE scalars; one of them may be NULL (= 0).  Same workspace as skgs_image_loss_backward. */
int skgs_image_loss_backward_terms(int32_t C, int32_t H, int32_t W, const float* pred, const float* gt,
    const int32_t* gt_index, const float* grad_l1, const float* grad_ssim, const void* workspace, size_t workspace_bytes,
    float* dL_dpred, skgs_stream_t stream);

/* ---- the two regularisers of stage `sp` on the LBS weights (value + gradient in one launch each) ----
 * Replace `loss_weight_sparsity` (networks/sk_gs.py:1339-1340: -mean(w log(w + eps) + (1 - w) log(1 - w + eps)) over the n = P K
 * weights) and `loss_weight_smooth` (:1357-1359: mean |w[i, k] - w[neighbours[i, g], k]| over i < P, g < G, k < K; neighbours [P, G]
 * int64 = `gs_knn_index`, the reference's 20-neighbour table of the Gaussians incl. the Gaussian itself) as the shipped configuration
 * runs them on outputs['_knn_w'] in EVERY stage-sp iteration (exps/default.yaml:85-86, sk_gs.py:1572-1574); in torch the second is a
 * [P, G, K] gather whose backward is a sort-based index_put (~1 ms at P = 1e5).  grad [n] / [P, K]: d value / d w (the caller multiplies
 * by the incoming cotangent); partials: skgs_weight_reg_partials() floats whose sum is the value (fixed-order partial sums). */
int32_t skgs_weight_reg_partials(void);
int skgs_weight_sparsity(int64_t n, const float* weights, float eps, float* grad, float* partials, skgs_stream_t stream);
/* inverse_offsets [P + 1] / inverse_sources [number of valid (i, g) pairs] (int32, both or neither): for every Gaussian j the Gaussians i
 * that list it, i.e. the pairs sorted by neighbours[i, g] -- built once per neighbour table (the reference rebuilds gs_knn_index every
 * 1000-3000 iterations).  With them the launch uses no atomics (the [P,K] table stays in L2); without them the gradient is summed by float
 * atomics (21 M of them at P = 1e5: 1.2 ms). */
int skgs_weight_smooth(int32_t P, int32_t K, int32_t G, const float* weights, const int64_t* neighbours, const int32_t* inverse_offsets,
    const int32_t* inverse_sources, float* grad, float* partials, skgs_stream_t stream);

/* ---- the live view slot from the reference's per-view tensors (no host read-back) ----
 * Replaces the host side of `prepare_inputs` (networks/gaussian_splatting.py:247-300) for one view: the reference builds
 * GaussianRasterizationSettings from `info` with `math.tan(0.5 * FoV[b, 0])` and `info['Tw2v']` on the host -- a blocking
 * device-to-host copy per iteration once `tensor_to(data, device)` (train.py:180) has put `info` on the GPU.  This launch writes
 * the 64-word record the kernels read their camera from (skgs_raster_inputs.viewmatrix / projmatrix / campos /
 * tanfov_device, skgs_bone_chain_*'s frame_index, the deform network's time; layout: sk_gs_amd/view_slot.py) from DEVICE
 * inputs:  viewmatrix = Tw2v^T, projmatrix = (Tv2c Tw2v)^T (:278-279), campos, tanfov = tan(0.5 FoV) (float product,
 * double tangent, as the host expression), time = *time, frame = *frame_index_device (int64, the data loader's `time_id`) or
 * the host value frame_index when that pointer is NULL.  Tw2v, Tv2c: [4,4] row-major; fov: [2]. */
int skgs_view_slot_fill(const float* Tw2v, const float* Tv2c, const float* campos, const float* fov, const float* time,
    const int64_t* frame_index_device, int32_t frame_index, int32_t target_index, float* slot, skgs_stream_t stream);

/* ---- multi-tensor Adam step in one launch (scope row (f)-2) ----
 * Replaces torch.optim.Adam(eps=1e-15) as configured by exps/default.yaml:122-125 over the parameter groups of
 * networks/gaussian_splatting.py:443-453 (amsgrad off, no weight decay).  `tensors` is a DEVICE array of descriptors
 *   struct { float* param; const float* grad; float* exp_avg; float* exp_avg_sq; int64_t n; int64_t chunk0; float lr;
 *            int32_t sched; }   (skgs_adam_tensor_bytes() = 56; sched: 0, or k > 0 = learning-rate schedule k - 1, see above)
 * with chunk0 = running sum of ceil(n / skgs_adam_chunk_elems()) and total_chunks the final sum.  step_state is the
 * optimizer's DEVICE state: skgs_adam_state_bytes() (256) bytes, 8-byte aligned, zero-initialised = no step taken.  Word
 * 0 is the number of steps taken so far as a float; doubles at bytes 8 and 16 hold 1 - beta1^steps and 1 - beta2^steps
 * (set them too when restoring a count: load_state_dict); the call advances all of it (hipGraph-capturable). */
/* ---- learning-rate schedules on the device ----
 * The reference calls update_learning_rate before EVERY training step (train.py:140-141): the `xyz` group follows
 * get_expon_lr_func (networks/gaussian_splatting.py:56-84,455-470: log-linear from lr_init to lr_final over max_steps, eased in by
 * lr_delay_mult + (1 - lr_delay_mult) sin(pi/2 clip(step / lr_delay_steps))), the deform networks' groups a second such schedule
 * with a stage offset (networks/sk_gs.py:611-632).  Up to 4 schedules live in the optimizer's device state: a tensor descriptor
 * whose `sched` word (the int32 after `lr`, 0 = none) is k > 0 takes schedule k - 1's rate instead of its own `lr`; the launch that
 * ADVANCES the step counter (skgs_adam_step, skgs_adam_step_range(advance = 1), skgs_adam_step_tail) evaluates every schedule for
 * the training step that follows -- in the double arithmetic numpy gives the reference, rounded to the float the update uses --
 * and keeps the closed step's rates for pieces that run after the advance but belong to it (after_advance).  No host in the loop:
 * a step replayed inside a hipGraph, several steps per replay, follows the reference's rate step for step.
 * Training step numbers are 1-based (the reference passes global_step + 1 / self._step); a schedule sees step - step_offset. */
typedef struct skgs_lr_schedule {
  double lr_init, lr_final, lr_delay_mult;
  double log_lr_init, log_lr_final; /* np.log of the two rates, formed by the host: the reference's own arithmetic (0 if the rate is 0) */
  int32_t lr_delay_steps; /* 0: no ease-in */
  int32_t max_steps;
  int32_t step_offset;    /* sk_gs.py:621-626: the stage's first step */
  int32_t reserved;
} skgs_lr_schedule;
/* schedules: DEVICE array of n <= 4 entries, alive and unchanged while the optimizer steps.  Evaluates them for the state's current
 * count (call it again after restoring a step count).  n = 0 removes them. */
int skgs_adam_set_lr_schedules(float* step_state, const skgs_lr_schedule* schedules, int32_t n, skgs_stream_t stream);
size_t skgs_adam_tensor_bytes(void);
size_t skgs_adam_state_bytes(void);
int64_t skgs_adam_chunk_elems(void);
int skgs_adam_step(int32_t n_tensors, const void* tensors, int64_t total_chunks, double beta1, double beta2, double eps,
    float* step_state, float* zero_after /* NULL, or zero_n floats cleared after the update (with the counter bump) */,
    int64_t zero_n, skgs_stream_t stream);
/* The same step taken in pieces: chunks [chunk_begin, chunk_end) of the table (whole tensors) get the update of step
 * *step_count + 1; only a piece with advance = 1 moves the counter and clears zero_after, and it must be ordered after
 * every other piece of that step.  Pieces may run on different streams beside backward kernels that do not touch their
 * tensors (the SH rows are final after the rasterizer backward, the Gaussian rows after the skinning backward: their
 * Adam runs beside the bone-chain / deform-network backward).  chunk_begin == chunk_end, advance = 1: counter only. */
int skgs_adam_step_range(int32_t n_tensors, const void* tensors, int64_t chunk_begin, int64_t chunk_end, double beta1,
    double beta2, double eps, float* step_state, int32_t advance, float* zero_after, int64_t zero_n, skgs_stream_t stream);
/* LBS_method 'W' (networks/sk_gs.py:469-471, exps/default.yaml:35): the Adam update of the dense [P, M] logit table WITHOUT the
 * dense gradient -- skgs_adam_step_range(advance = 0) for that one tensor, restricted to the 32-column tiles of a row that have
 * ever received a gradient (every other entry has g = m = v = 0: its update is exactly zero).  Bit-identical parameters and
 * moments to skgs_lbs_weights_backward + the dense step.  weights / indices / g_weights [P,K]: this step's softmax weights,
 * neighbours and the weights' cotangent; tensor: the table's descriptor in the optimizer's DEVICE table (skgs_adam_step:
 * param, exp_avg, exp_avg_sq, lr are read from it); tile_mask [P] uint32, persistent: zero at the start of training, else
 * skgs_adam_logit_mask_rebuild (a tile is live when any of its moments is non-zero).  M <= 1024, K <= 16. */
int skgs_adam_logit_rows(int32_t P, int32_t M, int32_t K, const float* weights, const int64_t* indices, const float* g_weights,
    const void* tensor, uint32_t* tile_mask, double beta1, double beta2, double eps, const float* step_state, int32_t after_advance,
    skgs_stream_t stream);
/* The sparse visit for an optimizer that KEEPS the dense gradient (torch.optim.Adam over the reference's own state tensors,
 * sk_gs_amd/reference_accel.py): the update of skgs_adam_step_range(advance = 0) for the ONE [P, M] tensor `tensor` describes, reading
 * its dense `grad`, restricted to the 32-column tiles in tile_mask [P] -- which first grows by the tiles of indices [P,K] (NULL / K = 0:
 * none known) and, with scan_gradient != 0, by every tile that holds a non-zero gradient (one pass over the gradient).  Exact: an
 * element whose gradient and moments are all zero does not move under Adam.  tile_mask must cover every non-zero moment
 * (skgs_adam_logit_mask_rebuild when it is first used).  M <= 1024. */
int skgs_adam_masked_rows(int32_t P, int32_t M, int32_t K, const int64_t* indices, int32_t scan_gradient, const void* tensor,
    uint32_t* tile_mask, double beta1, double beta2, double eps, const float* step_state, int32_t after_advance, skgs_stream_t stream);
int skgs_adam_logit_mask_rebuild(int32_t P, int32_t M, const float* exp_avg, const float* exp_avg_sq, uint32_t* tile_mask,
    skgs_stream_t stream);
/* The closing piece of a step (it advances the counter) whose range holds a tensor with an unfinished gradient: the
 * workgroup that owns chunk freq_chunk (the single chunk of that small tensor: the joint positions) first runs
 * skgs_freq_encode_backward(freq_B, freq_D, freq_degree, freq_grad_out, freq_out, freq_ld_out, freq_grad_x,
 * freq_accumulate) itself -- the gradient is completed and consumed without a launch in between.  freq_grad_x = NULL:
 * skgs_adam_step_range(..., advance = 1).
 * next_view (may be NULL): the last act of the launch puts the NEXT training view's record into the live view slot the
 * kernels read their camera / time / target index from (skgs_raster_inputs.tanfov_device, frame_index, gt_index):
 * slot[0..words) = table[order[*cursor % n_order]], *cursor += 1 -- a loop that walks its views in a known order then needs
 * no copy between two replays of its captured step. */
typedef struct skgs_view_advance {
  const void* table;       /* DEVICE [views][words] 32-bit words */
  const int32_t* order;    /* DEVICE [n_order] view indices */
  int32_t* cursor;         /* DEVICE counter; with n_order = 0 a pair {counter, length of order}: a captured graph then
                            * follows a new order of any length uploaded IN PLACE (same storage) between replays */
  void* slot;              /* DEVICE [words] */
  int32_t n_order, words;  /* n_order = 0: read the length from cursor[1] */
} skgs_view_advance;
int skgs_adam_step_tail(int32_t n_tensors, const void* tensors, int64_t chunk_begin, int64_t chunk_end, double beta1,
    double beta2, double eps, float* step_state, float* zero_after, int64_t zero_n, int64_t freq_chunk, int32_t freq_B,
    int32_t freq_D, int32_t freq_degree, const float* freq_grad_out, const float* freq_out, int32_t freq_ld_out,
    float* freq_grad_x, int32_t freq_accumulate, const skgs_view_advance* next_view, skgs_stream_t stream);

/* ---- bone-transform producer of the skeleton stage (scope row (f)-3) ----
 * SimpleDeformationNetwork (networks/sk_gs.py:134-164): FreqEncoder (my_ext/_C/src/nerf/freqencoder.cu:7-60) +
 * MLP_with_skips (my_ext/blocks/mlp.py:43-85), evaluated on one row per bone.  One launch per linear layer and direction:
 *   forward : Y[B,out] = act([X1 | X2] W^T + bias),  W [out, in1 + in2] (torch.nn.Linear layout), X2 optional (skip input)
 *   backward: gZ = gY * (Y > 0) when relu;  gW = gZ^T [X1 | X2];  gb = sum_b gZ;  gX1 = gZ W[:, :in1] (written);
 *             gX2 = gZ W[:, in1:]; accumulate_gx bit 0 / bit 1: add to gX1 / gX2 instead of writing.  gb, gX1, gX2 may be
 *             NULL.
 * freq encode: out[b, :] = [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(deg-1) x), cos(..)] grouped per frequency (C = D + 2 D deg
 * columns written at row stride ld_out). */
int skgs_freq_encode_forward(int32_t B, int32_t D, int32_t degree, const float* x, int32_t ld_x /* 0: one input row
    for all B output rows */, float* out, int32_t ld_out, skgs_stream_t stream);
int skgs_freq_encode_backward(int32_t B, int32_t D, int32_t degree, const float* grad_out, const float* out, int32_t ld_out,
    float* grad_x, int32_t accumulate /* 0: grad_x is written, 1: added to */, skgs_stream_t stream);
int skgs_linear_forward(int32_t B, int32_t in1, int32_t in2, int32_t out, const float* X1, int32_t ldx1, const float* X2,
    int32_t ldx2, const float* W, const float* bias, float* Y, int32_t ldy, int32_t relu, skgs_stream_t stream);
int skgs_linear_backward(int32_t B, int32_t in1, int32_t in2, int32_t out, const float* X1, int32_t ldx1, const float* X2,
    int32_t ldx2, const float* W, const float* Y, const float* gY, int32_t ldy, int32_t relu, float* gW, float* gb,
    float* gX1, int32_t ldg1, float* gX2, int32_t ldg2, int32_t accumulate_gx, skgs_stream_t stream);

/* ---- the same network as ONE persistent launch per direction (csrc/mlp_fused.hip) ----
 * Replaces the whole SimpleDeformationNetwork.forward call of kinematic() (networks/sk_gs.py:1073-1074; module :134-164,
 * MLP_with_skips my_ext/blocks/mlp.py:43-85, FreqEncoder my_ext/_C/src/nerf/freqencoder.cu:7-60) and its autograd
 * backward, for B <= 48 rows (one per bone), hidden width 256 (every reference config), encoded input a multiple of 4 and
 * <= 128 wide, at most 10 layers with the heads; 16-byte aligned weights.  Other shapes: the per-layer entry points above.
 * Layer l computes act([a_{l-1} | x0] W_l^T + b_l): in_hidden = 0 for the first layer and `hidden` afterwards, in_x0 = the
 * encoded width where the layer reads the encoded input (first layer, layers behind a skip) else 0; the last entry is the
 * heads stored as one matrix (relu = 0).  W is [out, in_hidden + in_x0] (torch.nn.Linear layout).
 *   forward : x0 [B, IN] (optional copy of the encoded input), acts [n_layers-1][B][hidden] (saved for the backward),
 *             out [B, out_last]
 *   backward: writes layer[l].gW / gb (gb may be NULL) and, if g_x0 != NULL, dL/dx0 [B, IN]
 * `t` is a DEVICE pointer to t_dim floats (the same time for every row).  The workspace
 * (skgs_deform_mlp_workspace_bytes, 256-byte aligned) must be prepared ONCE with skgs_deform_mlp_workspace_init before its
 * first use and then belongs to the library: it holds the in-launch exchange images and the launch counters, so consecutive
 * (also hipGraph-replayed) launches need no clearing.  skgs_deform_mlp_status copies four words: {forward launches, launches
 * that gave up waiting for another workgroup (must stay 0), diagnostics switch, backward launches}.  B <= 48 rows. */
#define SKGS_MLP_MAX_LAYERS 12
typedef struct skgs_mlp_layer {
  const float* W;
  const float* bias; /* may be NULL */
  float* gW;         /* backward only */
  float* gb;         /* backward only, may be NULL */
  int32_t in_hidden, in_x0, out, relu;
} skgs_mlp_layer;
typedef struct skgs_mlp_desc {
  int32_t B;                                 /* rows */
  int32_t p_dim, p_degree, t_dim, t_degree;  /* FreqEncoder(points) | FreqEncoder(t) */
  int32_t hidden, n_layers;                  /* n_layers counts the heads */
  skgs_mlp_layer layer[SKGS_MLP_MAX_LAYERS];
  /* optional: the last layer's columns as separate tensors (the reference's `last` ModuleList, mlp.py:76-83: sk_r |
   * d_rot | d_scale).  n_heads = 0: one [B, out] tensor (`out` / `g_out` arguments); else head j is [B, head_dim[j]] at
   * head_out[j] (forward) / head_gout[j] (backward) and the `out` / `g_out` arguments may be NULL. */
  int32_t n_heads, head_dim[4];
  float* head_out[4];
  const float* head_gout[4];
} skgs_mlp_desc;
size_t skgs_deform_mlp_workspace_bytes(const skgs_mlp_desc* d);
int skgs_deform_mlp_workspace_init(void* workspace, size_t workspace_bytes, skgs_stream_t stream);
/* Where the network's 32 workgroups run (process-wide; mode < 0: only ask; returns the mode in force before the call).
 *   0: blocks 0..31 of the launch (the dispatcher deals them four to each XCD); the in-launch exchange of the [B, hidden]
 *      activations goes through write-through (sc1) stores: every hop crosses the fabric.
 *   1 (default on a whole MI355X, 256 CUs; environment SKGS_MLP_XCD overrides): blocks 0, 8, .. 248 -- blocks of one residue mod 8
 *      are observed to share an XCD.  Every launch CHECKS that: each workgroup publishes the XCC_ID it runs on with its first
 *      slab, all read all 32 with their first gather, and only a launch found on ONE XCD sends the later slabs as plain stores
 *      (they stay in that XCD's L2: a hop is an L2 round trip, 0.76 us instead of 1.3); any other placement keeps the
 *      write-through stores.  Words 4 / 5 of the workspace header count the forward / backward launches that ran that way.
 *      Measured at config #1: skeleton forward 37.1 -> 34.4 us, skeleton backward (the rows' Adam on the other seven XCDs, none
 *      beside the network) 58.6 -> 53.4 us, the step 0.3388 -> 0.3294 ms.
 *   2: the placement of 1 with write-through stores throughout;  3: mode 1 with a falsified census (tests of the fall-back).
 * Mode 1 is for ONE fused network launch at a time per device (a training process): its 32 workgroups fill one XCD (one per CU),
 * so two such launches dispatched in the same microsecond to the same XCD would each hold CUs the other's missing workgroups
 * need, and both would give up after their bounded spins (status word 1; never a wrong result).  Processes that share a GPU and
 * run in lockstep (SKGS_SHARE_GPU=1) therefore default to mode 0; callers that launch from several streams at once choose 0 too.
 * No reference counterpart (the reference runs the network as ~65 torch launches, networks/sk_gs.py:1073-1074). */
int32_t skgs_deform_mlp_xcd_mode(int32_t mode);
int skgs_deform_mlp_forward(const skgs_mlp_desc* d, const float* points, const float* t, float* x0, float* acts, float* out,
    void* workspace, size_t workspace_bytes, skgs_stream_t stream);
/* x0: the [B, IN] encoded input the forward call wrote (its `x0` argument), or NULL: it is re-encoded from points / t
 * (B x IN sines per workgroup, ~3 us). */
int skgs_deform_mlp_backward(const skgs_mlp_desc* d, const float* points, const float* t, const float* x0, const float* acts,
    const float* g_out, float* g_x0, void* workspace, size_t workspace_bytes, skgs_stream_t stream);
/* The same launch with a side job: the network occupies 32 of the 256 CUs for ~30 us; the workgroups added for the other
 * CUs apply the Adam update (skgs_adam_step_range with advance = 0) of chunks [chunk_begin, chunk_end) of an optimizer
 * table -- parameters whose gradients are final before this call (the per-Gaussian rows, after the skinning backward) and
 * which this call neither reads nor writes.  side = NULL: skgs_deform_mlp_backward.  Close the step with the remaining
 * pieces and one advance (skgs_adam_step_range). */
typedef struct skgs_adam_range {
  int32_t n_tensors;
  const void* tensors;               /* device descriptor table, see skgs_adam_step */
  int64_t chunk_begin, chunk_end;
  double beta1, beta2, eps;
  const float* step_count;           /* the optimizer's device state (skgs_adam_state_bytes()): read, not advanced */
  int32_t after_advance;             /* 0: the piece precedes the launch that advances the counter (bias correction of step
                                        count + 1); 1: it follows it but belongs to that step (correction of step count):
                                        the piece that rides on the NEXT skeleton-forward launch */
} skgs_adam_range;
int skgs_deform_mlp_backward_adam(const skgs_mlp_desc* d, const float* points, const float* t, const float* x0,
    const float* acts, const float* g_out, float* g_x0, void* workspace, size_t workspace_bytes, const skgs_adam_range* side,
    skgs_stream_t stream);
/* ---- the skeleton stage as ONE launch per direction: network + kinematic chain ----
 * kinematic() (networks/sk_gs.py:1069-1107) = sk_deform_net -> normalised joint rotations -> chain of SE3 products.  The
 * network's rows are the bones (d->B == M) and its first head the raw rotations (d->head_dim[0] == 4): the workgroup
 * that owns those columns runs skgs_bone_chain_forward's body right after the heads (one launch and ~4 us less than the two
 * calls); in the backward every workgroup runs skgs_bone_chain_backward's body in its prologue, behind its weight loads --
 * the gradient of the raw rotations never goes through memory (d->head_gout[0], if not NULL, still receives a copy; the
 * other heads' gradients are read as usual). */
typedef struct skgs_bone_chain_desc {
  int32_t M, root, num_levels;
  const int32_t *parents, *level_nodes, *level_start;   /* as for skgs_bone_chain_forward */
  const float* joints;                                  /* [M,3] */
  const float* global_T;                                /* [7], or [frames,7] with frame_index; may be NULL */
  const int32_t* frame_index;                           /* DEVICE int32 or NULL */
  float* bone_T;                                        /* forward: [M,7] written */
  float* chain_A;                                       /* forward: [M,7] written (may be NULL); backward: read */
  const float* sk_r_raw;                                /* backward: the forward's head 0, [M,4] */
  const float* g_bone_T;                                /* backward: [M,7] */
  float* g_joints;                                      /* backward: [M,3] chain part, written; may be NULL */
  float* g_global_T;                                    /* backward: [7] (row frame_index), written; may be NULL */
  float* sk_cache;                                      /* skgs_skeleton_forward only, may be NULL: [frames, M, 11] -- row
                                                         * frame_index receives [normalised joint rotation | d_rot | d_scale],
                                                         * the reference's `self.sk_cache[time_id] = ...` under no_grad in
                                                         * every training step (networks/sk_gs.py:1077-1079), which the
                                                         * test-time path interpolates (:1080-1085) */
} skgs_bone_chain_desc;
int skgs_skeleton_forward(const skgs_mlp_desc* d, const skgs_bone_chain_desc* bones, const float* points, const float* t,
    float* x0, float* acts, void* workspace, size_t workspace_bytes, const skgs_adam_range* side /* may be NULL */,
    skgs_stream_t stream);
/* skgs_skeleton_backward follows the skgs_skeleton_forward of the SAME frame on the SAME workspace (as it must for `acts` and
 * `chain_A`): with a [frames,7] table it takes the frame's row of global_T from the copy that forward left in the workspace header
 * (words 8..14) instead of loading it through frame_index[0] again. */
int skgs_skeleton_backward(const skgs_mlp_desc* d, const skgs_bone_chain_desc* bones, const float* points, const float* t,
    const float* x0, const float* acts, float* g_x0, void* workspace, size_t workspace_bytes, const skgs_adam_range* side,
    skgs_stream_t stream);
int skgs_deform_mlp_status(const void* workspace, uint32_t* host_words4, skgs_stream_t stream);

/* ---- calc_LBS_weight of the SUPERPOINT stage (networks/sk_gs.py:751-774 as sp_stage calls it, :844) ----
 * K (<= 16) nearest of the M superpoints by squared L2 over [xyz | hyper_feature] (F = 8 hyper dimensions, or F = 0: xyz only;
 * pytorch3d.knn_points semantics: ascending, ties to the lower index; the positions carry no gradient, :753-755), then one of
 *   sp_W != NULL          : w = softmax_k(sp_W[p, idx])                                   (`W`, sk_gs.py:767-768)
 *   sp_radius_raw != NULL : w = (exp(-d / (2 r^2)) [* s] + 1e-7) / sum_k(...),  r = exp(_sp_radius), s = sigmoid(_sp_weight)
 *                                                                                         (`kernel` / `weighted_kernel`, :760-766)
 *   neither               : w = softmax_k(-d / temperature)                               (`dist`, :770)
 * in ONE launch.  points [P,3], feature [P,F], sp_points [M,3], sp_feature [M,F], sp_radius_raw / sp_weight_raw [M] (the RAW
 * parameters: the kernels apply exp / sigmoid, sk_gs.py:547-553), sp_W [P,M].  out_idx [P,K] int64, out_weights [P,K],
 * out_dist [P,K] (may be NULL for `W`).  Same arithmetic as skgs_knn_dist_weights_forward / skgs_lbs_weights_forward on the
 * concatenated rows (bit-identical weights).
 * params_activated != 0: sp_radius_raw / sp_weight_raw hold the ACTIVATED values r = exp(_sp_radius), s = sigmoid(_sp_weight) --
 * what the reference's calc_LBS_weight receives (sk_gs.py:759-766), the operator path (sk_gs_amd.deform.calc_lbs_weight) --
 * and the backward returns the gradients w.r.t. those.
 * Backward of the two distance-based weightings: g_weights [P,K] -> g_feature [P,F] (written), g_sp_feature [M,F],
 * g_sp_radius [M], g_sp_weight [M] w.r.t. the raw parameters (written; any may be NULL).  workspace:
 * skgs_sp_lbs_weights_workspace_bytes(P, M, F).  (`W`: skgs_lbs_weights_backward.)
 * sp_order (may be NULL): a permutation of 0..M-1, the order the superpoints are SCANNED in.  It cannot change the result (the
 * list is kept by (distance, id)); a spatial order (Z-order of sp_points) makes the candidates a wave looks at together
 * neighbours, so its wave-wide early-outs skip most of them.  sp_rank (may be NULL; with sp_order): its inverse, rank[id] = the
 * scan position of superpoint id.  With it every wave starts its scan at the superpoint that was nearest to its first Gaussian
 * in the PREVIOUS call with the same out_idx buffer (read before it is overwritten; any content is a valid hint) and walks
 * outwards: the lists fill with near neighbours first and the rest of the scan is skipped wave-wide.
 * pairs (skgs_sp_pairs_bytes(P, M, K) bytes, may be NULL): the forward also files every (Gaussian, neighbour) pair under its
 * superpoint -- inverse lists that skgs_sp_skinning_backward walks; the call clears them first.  A superpoint whose list
 * outgrows its capacity (16 x the mean list, >= 4096 entries) sets the overflow word (byte 4 of the buffer; cleared by the next
 * forward's preparation) and counts the forward in the overflow-EVENT word (byte 8), which the library never clears: the pairs
 * beyond the capacity are missing from the backward's sums, and a caller that looks only every N steps must still learn of it. */
int skgs_sp_lbs_weights_forward(int32_t P, int32_t M, int32_t K, int32_t F, const float* points, const float* feature,
    const float* sp_points, const float* sp_feature, const float* sp_radius_raw, const float* sp_weight_raw, float temperature,
    const float* sp_W, const int32_t* sp_order /* or NULL */, const int32_t* sp_rank /* or NULL */, int64_t* out_idx,
    float* out_weights, float* out_dist, void* pairs /* or NULL */, size_t pairs_bytes, int32_t params_activated,
    int32_t pairs_prepared /* the table and counters of `pairs` were prepared by skgs_sp_net_forward's launch (skgs_sp_prepare) */,
    skgs_stream_t stream);
size_t skgs_sp_lbs_weights_workspace_bytes(int32_t P, int32_t M, int32_t F);
int skgs_sp_lbs_weights_backward(int32_t P, int32_t M, int32_t K, int32_t F, const float* feature, const float* sp_feature,
    const float* sp_radius_raw, const float* sp_weight_raw, float temperature, const float* weights, const int64_t* indices,
    const float* nn_dist, const float* g_weights, float* g_feature, float* g_sp_feature, float* g_sp_radius, float* g_sp_weight,
    void* workspace, size_t workspace_bytes, int32_t params_activated, skgs_stream_t stream);

/* ---- skinning + weighting backward of the SUPERPOINT stage in one call, no atomics on the superpoint tables ----
 * = skgs_lbs_deform_backward + skgs_sp_lbs_weights_backward for M in the hundreds (their LDS float atomics are LDS-bound there:
 * 58 + 68 us at P = 1e5, M = 512).  Three launches: rows (one lane per Gaussian: its parameter gradients, g_weights, the
 * weighting's chain rule, hyper_feature.grad, and a compact payload), bones (4 workgroups per superpoint walk the inverse list
 * the forward filed and tree-reduce the 24 per-superpoint sums), finalize (adds the slices, d exp / d sigmoid of the raw
 * radius / weight parameters).  `in`: as skgs_lbs_deform_backward (bone_T = spT [M,7], bone_drot = the unit quaternion,
 * bone_dscale = d_scaling; weights / indices of the forward).  logit_weighting != 0: the `W` weighting (the weights do not
 * depend on the distances: only g_weights [P,K] leaves, feed it to skgs_lbs_weights_backward); otherwise nn_dist [P,K] and the
 * weighting's parameters as given to the forward.  Outputs are WRITTEN: g_xyz, g_log_scale, g_rot, g_opacity_logit [P,.],
 * g_feature [P,F], g_weights [P,K] (may be NULL), g_bone_T [M,7], g_bone_drot [M,4], g_bone_dscale [M,3], g_sp_feature [M,F],
 * g_sp_radius [M], g_sp_weight [M] (the last four may be NULL).  workspace: skgs_sp_skinning_backward_workspace_bytes. */
size_t skgs_sp_pairs_bytes(int32_t P, int32_t M, int32_t K);
size_t skgs_sp_skinning_backward_workspace_bytes(int32_t P, int32_t M, int32_t K);
/* The call below as a job of skgs_rasterize_backward (skgs_raster_grads.sp_skinning_job): its arguments without the four upstream
 * gradients -- the per-Gaussian launch that produces dL/d(means3D, scales, rotations, opacity) runs the ROWS pass on them in
 * registers (they are still written to the skgs_raster_grads outputs), the bones and finalize launches follow inside
 * skgs_rasterize_backward.  One launch less, the same arithmetic on the same values. */
typedef struct skgs_sp_skinning_job {
  const skgs_deform_inputs* in;
  int32_t F;
  const float *feature, *sp_feature, *sp_radius_raw, *sp_weight_raw;
  float temperature;
  int32_t logit_weighting;
  const float* nn_dist;
  float *g_weights, *g_xyz, *g_log_scale, *g_rot, *g_opacity_logit, *g_feature, *g_bone_T, *g_bone_drot, *g_bone_dscale, *g_sp_feature,
      *g_sp_radius, *g_sp_weight;
  void* pairs;
  size_t pairs_bytes;
  void* workspace;
  size_t workspace_bytes;
  const float* g_weights_extra; /* NULL, or [P,K]: a cotangent on the LBS weights from OUTSIDE the skinning -- the reference's loss
                                 * reads outputs['_knn_w'] in stage `sp` (`sparse`, `smooth`: networks/sk_gs.py:1339-1359,1572-1574) --
                                 * added to the skinning's own before the weighting's chain rule (and into g_weights) */
} skgs_sp_skinning_job;
int skgs_sp_skinning_backward(const skgs_deform_inputs* in, int32_t F, const float* feature, const float* sp_feature,
    const float* sp_radius_raw, const float* sp_weight_raw, float temperature, int32_t logit_weighting, const float* nn_dist,
    const float* g_means, const float* g_scales, const float* g_rotations, const float* g_opacity, float* g_weights, float* g_xyz,
    float* g_log_scale, float* g_rot, float* g_opacity_logit, float* g_feature, float* g_bone_T, float* g_bone_drot,
    float* g_bone_dscale, float* g_sp_feature, float* g_sp_radius, float* g_sp_weight, void* pairs, size_t pairs_bytes,
    void* workspace, size_t workspace_bytes, skgs_stream_t stream);

/* ---- the deform network of the SUPERPOINT stage (stage `sp`, networks/sk_gs.py:830-856) ----
 * sp_deform_net = DeformNetwork (networks/sk_gs.py:209-315) as the shipped configs build it (exps/default.yaml:4-11,31:
 * is_blender, D = 8, W = 256, skips = [4], position degree 10, time degree 6), evaluated on the M superpoints (512,
 * exps/default.yaml:25) every step:
 *   t_emb = time_w2 relu(time_w1 freq(t, 6) + time_b1) + time_b2          (timenet: Linear(13,256) ReLU Linear(256,30), :250-253)
 *   h = [freq(x, 10) | t_emb] (93);  8 x  h = relu(W[i] h + b[i]);  after layer 4  h = [freq(x) | t_emb | h]      (:300-306)
 *   raw = [warp(h) 3 | rotation(h) 4 | scaling(h) 3]                                                              (:308-310)
 * and the stage's epilogue (sk_gs.py:847, warp() :796-821): bone_T [M,7] = [d_xyz | u], d_rot [M,4] = u, d_scale [M,3] =
 * scaling, u = normalize(rotation + [0,0,0,1]) -- the three inputs of skgs_lbs_deform_forward.  All matrices in
 * torch.nn.Linear layout [out, in], contiguous: W[0] [256,93], W[5] [256,349] (input columns first), the others [256,256].
 * Two launches forward (a transposed copy of the hidden layers' weights into `saved`, then MFMA row blocks: 4 superpoints per
 * workgroup through the whole network, the weights streamed from that copy); two backward (row blocks, then all weight
 * gradients on the whole chip).  The backward WRITES the gradient of every parameter into the matching
 * pointer of `grads` (same struct; time ignored, points: see SKGS_SP_NET_LBS_C).  Cotangents: either g_raw [M,10] ([M,14] with the
 * local-rotation head; w.r.t. the raw row above) or
 * g_bone_T [M,7] / g_d_rot [M,4] / g_d_scale [M,3] (any may be NULL; the normalisation's backward runs in the launch).
 * No gradient w.r.t. points or time (the reference detaches the positions, sk_gs.py:746-748,845).
 * saved: skgs_sp_net_saved_bytes(M), written by the forward, read by the backward.  workspace:
 * skgs_sp_net_workspace_bytes(M), ZERO before the first call (the library keeps its first 256 bytes zero between calls).
 * side (may be NULL): an optimizer piece for the CUs the two backward launches leave idle (split between them), as
 * skgs_skeleton_backward. */
#define SKGS_SP_NET_LBS_C 1 /* skgs_sp_net.flags: warp_method LBS_c (exps/d_nerf_sc_gs.yaml:32, sk_gs.py:803-804, 1121-1122): the
                             * superpoint's translation is re-centred on the superpoint, bone_T = [d_xyz + x + R(u)(-x) | u]; the
                             * backward then also returns d loss / d sp_points through `grads->points` [M,3] (written; may be NULL) */
/* skgs_sp_net.flags: DeformNetwork(is_blender=False) (the class default, sk_gs.py:220; the HyperNeRF-style variant: no shipped
 * YAML selects it): NO time network -- t_emb = freq(t, degree) itself (sk_gs.py:297-301 without :299), so the encoded input has
 * 63 + 1 + 2 degree columns: W[0] [256, 64 + 2 degree], W[5] [256, 64 + 2 degree + 256]; time_w1 .. time_b2 are ignored (NULL).
 * degree <= 15.  The caller adds the stage's time noise (sk_gs.py:837-839) to `time` before the call. */
#define SKGS_SP_NET_RAW_TIME_FLAG 2
#define SKGS_SP_NET_RAW_TIME SKGS_SP_NET_RAW_TIME_FLAG
#define SKGS_SP_NET_RAW_TIME_DEGREE(degree) (SKGS_SP_NET_RAW_TIME_FLAG | ((degree) << 8))
typedef struct skgs_sp_net {
  int32_t M, flags;
  const float* points;                                   /* [M,3] */
  const float* time;                                     /* DEVICE scalar */
  const float *time_w1, *time_b1, *time_w2, *time_b2;    /* [256,13], [256], [30,256], [30] */
  const float* W[8];
  const float* b[8];
  const float *warp_w, *warp_b, *scaling_w, *scaling_b, *rotation_w, *rotation_b;   /* [3,256] [3] [3,256] [3] [4,256] [4] */
  /* sep_rot (DeformNetwork.local_rotation, sk_gs.py:275-282,315; both NULL: off): a fourth head [4,256] [4] whose output
   * g_rotation is what `warp` blends per Gaussian instead of d_rotation (sk_gs.py:848,818-821): d_rot = normalize(g_rotation +
   * [0,0,0,1]) while bone_T keeps normalize(d_rotation + [0,0,0,1]); the raw row is then [M,14] = [d_xyz | d_rotation | d_scaling |
   * g_rotation] */
  const float *local_w, *local_b;
} skgs_sp_net;
size_t skgs_sp_net_saved_bytes(int32_t M);
size_t skgs_sp_net_workspace_bytes(int32_t M);
/* prepare (may be NULL): the per-step preparation of the stage's SEARCH -- its superpoint table packed in scan order, the
 * inverse lists' header and counters cleared (what skgs_sp_lbs_weights_forward otherwise does in a small launch of its own) --
 * done by extra workgroups of this call's weight-transposition launch: the two are independent preparations of parameters.
 * Give the SAME pairs buffer / sp_order / sp_points / sp_feature to the skgs_sp_lbs_weights_forward call that follows, with
 * pairs_prepared = 1. */
typedef struct skgs_sp_prepare {
  int32_t P, M, K, F;
  const float* sp_points;
  const float* sp_feature;   /* [M,F] or NULL (F = 0) */
  const int32_t* sp_order;   /* or NULL */
  void* pairs;
  size_t pairs_bytes;
} skgs_sp_prepare;
int skgs_sp_net_forward(const skgs_sp_net* net, float* raw /* [M,10] ([M,14] with local_w) or NULL */, float* bone_T, float* d_rot, float* d_scale,
    void* saved, size_t saved_bytes, const skgs_sp_prepare* prepare, skgs_stream_t stream);
int skgs_sp_net_backward(const skgs_sp_net* net, const skgs_sp_net* grads, const float* g_bone_T, const float* g_d_rot,
    const float* g_d_scale, const float* g_raw, const void* saved, size_t saved_bytes, void* workspace, size_t workspace_bytes,
    const skgs_adam_range* side, skgs_stream_t stream);

/* ---- densification statistics of one training view (scope row (f)-4) ----
 * networks/sk_gs.py:1990-1997 + networks/gaussian_splatting.py:503-513: for every Gaussian with radii > 0
 *   max_radii2D = max(max_radii2D, radii); xyz_gradient_accum += |grad_means2D[:, :2]|; denom += 1.
 * grad_means2D [P,3] is dL_dmeans2D of skgs_rasterize_backward (`viewspace_points.grad`); accum, denom [P,1].
 * grad_multiplier scales the norm before it is accumulated: view-parallel training seeds its backward with 1 / world (so
 * that the gradients arrive pre-averaged), the statistic must see the UNSCALED gradient -- pass world there, 1 otherwise. */
int skgs_densify_stats(int32_t P, const int32_t* radii, const float* grad_means2D, float grad_multiplier,
    float* xyz_gradient_accum, float* denom, float* max_radii2D, skgs_stream_t stream);

/* ---- simple_knn (initialisation, not on the per-frame path): networks/gaussian_splatting.py:211-213 resolves `simple_knn` from the
 * extension for create_from_pcd (my_ext/_C/src/other/knn.cu:192-205): mean_dist2[p] = mean of the squared distances from point p
 * to its three nearest other points.  points [P,3], mean_dist2 [P]. */
int skgs_simple_knn(int32_t P, const float* points, float* mean_dist2, skgs_stream_t stream);

/* ---- densification surgery in one launch (scope row (f)-4) ----
 * Replaces the per-tensor indexing / concatenation of change_optimizer, prune_points and densification_postfix
 * (networks/gaussian_splatting.py:515-587) over the per-Gaussian parameters and their Adam moments: for every tensor t of
 * the DEVICE descriptor array `tensors` (struct { const float* src; float* dst; int32 row_floats; int32 fresh_is_zero; },
 * skgs_row_tensor_bytes() = 24)
 *     dst_t[i, :] = src_t[rows[i], :]                          i <  n_keep   (survivors)
 *     dst_t[i, :] = fresh_is_zero ? 0 : src_t[rows[i], :]      i >= n_keep   (new Gaussians: parameters copied from their
 *                                                                             parent, moments zero)
 * rows: device int64 [n_out]; max_row_floats: the largest row_floats of the table (sizes the grid). */
size_t skgs_row_tensor_bytes(void);
int skgs_gather_rows(int32_t n_tensors, const void* tensors, int64_t n_out, int64_t n_keep, const int64_t* rows,
    int32_t max_row_floats, skgs_stream_t stream);

/* ---- the decisions of adaptive density control on the device (networks/gaussian_splatting.py:589-650) ----
 * skgs_densify_select: clone / split masks (mean screen-space gradient accum / denom against max_grad, largest scale
 * against scene_extent) and the row list of the ONE gather that realises clone + split: rows = [not split] ++ [clone] ++
 * [split] x N; counts (DEVICE int32[3]) = the group sizes -- read them back to size the new tensors (n_keep = counts[0],
 * n_out = counts[0] + counts[1] + N counts[2]).  rows: capacity (2 + N) P int64; flags_ws: skgs_select_workspace_bytes(P)
 * bytes of scratch (16-byte aligned).
 * skgs_prune_select: rows = the survivors of prune() (opacity below min_opacity; with max_radii2D != NULL also screen radius
 * above max_screen_size or largest scale above world_size_limit), counts (DEVICE int32[1]) their number.
 * skgs_split_children: the n gathered children of split Gaussians (copies of their parents), in place: position +=
 * R(rot) (normals * scale), scale /= 0.8 N. */
size_t skgs_select_workspace_bytes(int32_t P);
int skgs_densify_select(int32_t P, const float* xyz_gradient_accum, const float* denom, const float* log_scale, float max_grad,
    float scene_extent, int32_t N, int64_t* rows, int32_t* counts, uint8_t* flags_ws, skgs_stream_t stream);
int skgs_prune_select(int32_t P, const float* opacity_logit, const float* max_radii2D, const float* log_scale,
    float min_opacity, float max_screen_size, float world_size_limit, int64_t* rows, int32_t* counts, uint8_t* flags_ws,
    skgs_stream_t stream);
int skgs_split_children(int32_t n, int32_t N, const float* normals, float* xyz, float* log_scale, const float* rot,
    skgs_stream_t stream);


/* Tuning knob of the blend kernels: pixels handled per lane (1, 2 or 4); 0 = heuristic on the tile count. */
void skgs_set_pixels_per_lane(int ppl);
/* Parity-test switch: blend kernels without FMA contraction, in the oracle's operation order, reproducible exp. */
void skgs_set_strict_math(int on);
/* Order in which the blend kernels' workgroups walk the tiles: 1 (default) groups of 8 tiles heaviest first (ranked inside
 * the sort launch from the per-tile counts), 0 raster order.  Results do not depend on it; for A/B timing (set it before
 * a hipGraph capture: the mode is a launch argument). */
void skgs_set_tile_order(int mode);

/* Per-kernel timing with HIP events recorded on the launch stream (bit k of the mask enables kernel id k; ids are
 * 0..skgs_profile_kernel_count()-1, names via skgs_profile_kernel_name). skgs_profile_collect waits for the events
 * recorded since the previous collect and returns their summed duration [ms] and the number of launches. */
void skgs_profile_enable(uint32_t kernel_mask);
int skgs_profile_kernel_count(void);
const char* skgs_profile_kernel_name(int kernel_id);
int skgs_profile_collect(int kernel_id, double* total_ms, int32_t* launches);

int skgs_fused_lbs_max_bones(void); /* = SKGS_FUSED_LBS_MAX_BONES of the loaded library */
const char* skgs_last_error(void);
int skgs_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SKGS_H_ */
