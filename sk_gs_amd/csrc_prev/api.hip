// api.hip -- C ABI of libskgs_hip.so (declared in include/skgs.h): argument checks, buffer carving, launch order.
#include <stdarg.h>
#include <stdio.h>

#include <algorithm>

#include "skgs_common.h"

namespace skgs {
static thread_local char g_err[512] = "";
int set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return 1;
}

// ---- fill kernel: used instead of hipMemsetAsync so that every node of a captured hipGraph is a plain kernel node
__global__ void fill_u32_kernel(uint32_t* __restrict__ p, uint32_t v, size_t n) {
  const size_t stride = (size_t) gridDim.x * blockDim.x;
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}
int fill_u32(void* p, uint32_t v, size_t n_words, hipStream_t s) {
  if (n_words == 0) return 0;
  const int blocks = (int) std::min<size_t>((n_words + 255) / 256, 2048);
  hipLaunchKernelGGL(fill_u32_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<uint32_t*>(p), v, n_words);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- the live view slot from the reference's per-view `info` tensors (skgs_view_slot_fill) -------------------------------
// One thread per output word; every input is a DEVICE load: no host read-back between the data loader and the launches.
__global__ void view_slot_fill_kernel(const float* __restrict__ Tw2v, const float* __restrict__ Tv2c,
    const float* __restrict__ campos, const float* __restrict__ fov, const float* __restrict__ time,
    const int64_t* __restrict__ frame64, int32_t frame_host, int32_t target_index, float* __restrict__ slot) {
#pragma clang fp contract(off)
  const int w = threadIdx.x;
  if (w < 16) {  // viewmatrix = Tw2v^T, row-major: word[4 i + j] = Tw2v[j][i]   (gaussian_splatting.py:278)
    slot[w] = Tw2v[4 * (w & 3) + (w >> 2)];
  } else if (w < 32) {  // projmatrix = (Tv2c Tw2v)^T: word[16 + 4 i + j] = sum_k Tv2c[j][k] Tw2v[k][i]   (:279)
    const int i = (w - 16) >> 2, j = (w - 16) & 3;
    float acc = Tv2c[4 * j + 0] * Tw2v[0 + i];
    acc = acc + Tv2c[4 * j + 1] * Tw2v[4 + i];
    acc = acc + Tv2c[4 * j + 2] * Tw2v[8 + i];
    acc = acc + Tv2c[4 * j + 3] * Tw2v[12 + i];
    slot[w] = acc;
  } else if (w < 35) {
    slot[w] = campos[w - 32];
  } else if (w == 36 || w == 37) {  // math.tan(0.5 * FoV[b, c]): the product in float32 (a tensor op), the tangent in double (:274-275)
    const float h = 0.5f * fov[w - 36];
    slot[w] = (float) tan((double) h);
  } else if (w == 38) {
    slot[w] = time ? time[0] : 0.f;
  } else if (w == 39) {
    reinterpret_cast<int32_t*>(slot)[w] = frame64 ? (int32_t) frame64[0] : frame_host;
  } else if (w == 40) {
    reinterpret_cast<int32_t*>(slot)[w] = target_index;
  }
}

// ---- per-kernel event timing ---------------------------------------------------------------------------------
namespace {
constexpr int PROF_RING = 4096;
struct ProfSlot {
  hipEvent_t start[PROF_RING], stop[PROF_RING];
  bool created = false;
  int used     = 0;     // pairs recorded since the last collect
  bool open    = false; // begin recorded, end pending
};
ProfSlot g_prof[K_COUNT];
uint32_t g_prof_mask = 0;
const char* const g_prof_names[K_COUNT] = {"preprocess_forward", "scan_tiles", "scatter", "tile_sort", "render_forward",
    "render_backward", "preprocess_backward", "deform_forward", "deform_backward", "knn_bones", "image_loss_forward",
    "image_loss_backward", "skeleton_forward", "skeleton_backward", "adam", "sp_net_forward", "sp_net_backward",
    "sp_knn_weights", "sp_knn_weights_backward"};
}  // namespace

void prof_begin(int kid, hipStream_t s) {
  if (!((g_prof_mask >> kid) & 1u)) return;
  ProfSlot& p = g_prof[kid];
  if (!p.created) {
    for (int i = 0; i < PROF_RING; ++i) {
      if (hipEventCreate(&p.start[i]) != hipSuccess || hipEventCreate(&p.stop[i]) != hipSuccess) return;
    }
    p.created = true;
  }
  if (p.used >= PROF_RING) return;  // ring full: stop sampling until collected
  p.open = hipEventRecord(p.start[p.used], s) == hipSuccess;
}
void prof_end(int kid, hipStream_t s) {
  if (!((g_prof_mask >> kid) & 1u)) return;
  ProfSlot& p = g_prof[kid];
  if (!p.open) return;
  if (hipEventRecord(p.stop[p.used], s) == hipSuccess) p.used++;
  p.open = false;
}

static int check_inputs(const skgs_raster_inputs* in) {
  SKGS_REQUIRE(in != nullptr, "inputs struct is NULL");
  SKGS_REQUIRE(in->P >= 0, "P must be >= 0");
  SKGS_REQUIRE(in->image_width > 0 && in->image_height > 0, "image size must be positive");
  if (in->P == 0) return 0;
  SKGS_REQUIRE(in->means3D != nullptr, "means3D must have dimensions (num_points, 3)");
  SKGS_REQUIRE(in->opacity != nullptr, "opacity is required");
  SKGS_REQUIRE(in->viewmatrix && in->projmatrix && in->campos, "viewmatrix / projmatrix / campos are required");
  SKGS_REQUIRE((in->sh != nullptr) != (in->colors_precomp != nullptr),
      "colour input: pass either SH coefficients or precomputed colours, not both and not neither");
  SKGS_REQUIRE(((in->scales != nullptr) && (in->rotations != nullptr)) != (in->cov3D_precomp != nullptr),
      "shape input: pass either (scales, rotations) or a precomputed 3D covariance, not both and not neither");
  if (in->sh) {
    SKGS_REQUIRE(in->sh_degree >= 0 && in->sh_degree <= 3, "sh_degree must be in [0,3]");
    SKGS_REQUIRE(in->sh_coeffs >= (in->sh_degree + 1) * (in->sh_degree + 1), "sh has too few coefficients for sh_degree");
    SKGS_REQUIRE(!in->sh_rest || in->sh_coeffs >= 2, "sh_rest given but sh_coeffs < 2");
  } else {
    SKGS_REQUIRE(!in->sh_rest, "sh_rest given without sh");
  }
  SKGS_REQUIRE(in->E >= 0 && in->E <= SKGS_MAX_RENDER_EXTRA, "Only Support 0,1,2,3,4 extra features");
  return 0;
}
static int check_buffers(const skgs_raster_inputs* in, const skgs_raster_buffers* buf, bool need_binning) {
  SKGS_REQUIRE(buf != nullptr && buf->geom && buf->img, "geom / img buffers are required");
  SKGS_REQUIRE(buf->geom_bytes >= geom_bytes(in->P), "geom buffer too small");
  SKGS_REQUIRE(buf->img_bytes >= img_bytes(in->image_width, in->image_height), "img buffer too small");
  if (need_binning) SKGS_REQUIRE(buf->binning != nullptr || buf->binning_bytes == 0, "binning buffer is NULL");
  return 0;
}
}  // namespace skgs

using namespace skgs;

extern "C" {

int skgs_view_slot_fill(const float* Tw2v, const float* Tv2c, const float* campos, const float* fov, const float* time,
    const int64_t* frame_index_device, int32_t frame_index, int32_t target_index, float* slot, skgs_stream_t stream) {
  SKGS_REQUIRE(Tw2v && Tv2c && campos && fov && slot, "view_slot_fill: bad argument");
  hipLaunchKernelGGL(view_slot_fill_kernel, dim3(1), dim3(64), 0, (hipStream_t) stream, Tw2v, Tv2c, campos, fov, time,
      frame_index_device, frame_index, target_index, slot);
  SKGS_CHECK_HIP(hipGetLastError());
  return 0;
}

void skgs_profile_enable(uint32_t kernel_mask) { g_prof_mask = kernel_mask; }
int skgs_profile_kernel_count(void) { return K_COUNT; }
const char* skgs_profile_kernel_name(int kid) { return (kid >= 0 && kid < K_COUNT) ? g_prof_names[kid] : ""; }
/* Waits for the recorded events of kernel `kid`, returns their summed duration and count, and resets the ring. */
int skgs_profile_collect(int kid, double* total_ms, int32_t* launches) {
  SKGS_REQUIRE(kid >= 0 && kid < K_COUNT && total_ms && launches, "profile_collect: bad argument");
  ProfSlot& p = g_prof[kid];
  double sum  = 0.0;
  for (int i = 0; i < p.used; ++i) {
    SKGS_CHECK_HIP(hipEventSynchronize(p.stop[i]));
    float ms = 0.f;
    SKGS_CHECK_HIP(hipEventElapsedTime(&ms, p.start[i], p.stop[i]));
    sum += ms;
  }
  *total_ms = sum;
  *launches = p.used;
  p.used    = 0;
  return 0;
}

int skgs_fused_lbs_max_bones(void) { return SKGS_FUSED_LBS_MAX_BONES; }
const char* skgs_last_error(void) { return g_err; }
int skgs_version(void) { return SKGS_VERSION; }

size_t skgs_geom_buffer_bytes(int32_t P) { return geom_bytes(P); }
size_t skgs_img_buffer_bytes(int32_t W, int32_t H) { return img_bytes(W, H); }
size_t skgs_binning_buffer_bytes(int64_t capacity) { return bin_bytes(capacity < 0 ? 0 : capacity); }
int64_t skgs_binning_capacity(size_t bytes) { return bin_capacity(bytes); }
size_t skgs_backward_workspace_bytes(int32_t P) { return (size_t) P * GRAD_ROW * 4 + 256; }

int skgs_rasterize_forward_stage1(const skgs_raster_inputs* in, const skgs_raster_buffers* buf, int32_t* radii,
    int32_t* host_num_rendered, skgs_stream_t stream) {
  if (check_inputs(in) || check_buffers(in, buf, false)) return 1;
  SKGS_REQUIRE(in->tile_bucket_capacity <= 0, "tile_bucket_capacity: use skgs_rasterize_forward (no two-stage form)");
  SKGS_REQUIRE(radii != nullptr || in->P == 0, "radii output is required");
  hipStream_t s = (hipStream_t) stream;
  GeomView g    = geom_view(buf->geom);
  ImgView im    = img_view(buf->img, in->image_width, in->image_height);
  if (launch_preprocess_forward(*in, g, im, radii, s)) return 1;
  if (launch_scan_tiles(g, im, in->P, s)) return 1;
  if (host_num_rendered)
    SKGS_CHECK_HIP(hipMemcpyAsync(host_num_rendered, &g.hdr->num_rendered,
        sizeof(int32_t) * (in->host_status_words == 3 ? 3 : 1), hipMemcpyDeviceToHost, s));
  return 0;
}

int skgs_rasterize_forward_stage2(const skgs_raster_inputs* in, const skgs_raster_buffers* buf, float* out_color,
    float* out_opacity, float* out_extra, skgs_stream_t stream) {
  if (check_inputs(in) || check_buffers(in, buf, true)) return 1;
  SKGS_REQUIRE(out_color && out_opacity, "out_color / out_opacity are required");
  SKGS_REQUIRE(!(in->extras && in->E > 0) || out_extra, "out_extra is required when extras are given");
  hipStream_t s = (hipStream_t) stream;
  GeomView g    = geom_view(buf->geom);
  ImgView im    = img_view(buf->img, in->image_width, in->image_height);
  BinView b     = bin_view(buf->binning, buf->binning_bytes);
  if (launch_scatter_sort(*in, g, im, b, s)) return 1;
  return launch_render_forward(*in, g, im, b, out_color, out_opacity, out_extra, s);
}

int skgs_rasterize_forward(const skgs_raster_inputs* in, const skgs_raster_buffers* buf, int32_t* radii, float* out_color,
    float* out_opacity, float* out_extra, int32_t* host_num_rendered, skgs_stream_t stream) {
  if (in && in->tile_bucket_capacity > 0) {
    // bucket layout: preprocess (clears the per-tile cursors) -> scatter into the tiles' fixed slots -> sort -> blend
    if (check_inputs(in) || check_buffers(in, buf, true)) return 1;
    SKGS_REQUIRE(radii != nullptr || in->P == 0, "radii output is required");
    SKGS_REQUIRE(out_color && out_opacity, "out_color / out_opacity are required");
    SKGS_REQUIRE(!(in->extras && in->E > 0) || out_extra, "out_extra is required when extras are given");
    SKGS_REQUIRE(host_num_rendered == nullptr, "the bucket layout does not compute num_rendered");
    hipStream_t s = (hipStream_t) stream;
    GeomView g    = geom_view(buf->geom);
    ImgView im    = img_view(buf->img, in->image_width, in->image_height);
    BinView b     = bin_view(buf->binning, buf->binning_bytes);
    SKGS_REQUIRE(b.capacity >= (int64_t) im.T * in->tile_bucket_capacity,
        "binning buffer too small for tiles x tile_bucket_capacity instances");
    SKGS_REQUIRE((int64_t) im.T * in->tile_bucket_capacity < (int64_t) 1 << 32, "tiles x tile_bucket_capacity must fit 32 bits");
    if (launch_preprocess_forward(*in, g, im, radii, s)) return 1;
    if (launch_scatter_sort(*in, g, im, b, s)) return 1;
    return launch_render_forward(*in, g, im, b, out_color, out_opacity, out_extra, s);
  }
  if (skgs_rasterize_forward_stage1(in, buf, radii, host_num_rendered, stream)) return 1;
  return skgs_rasterize_forward_stage2(in, buf, out_color, out_opacity, out_extra, stream);
}

int skgs_read_status(const skgs_raster_buffers* buf, skgs_status* host_status, skgs_stream_t stream) {
  SKGS_REQUIRE(buf && buf->geom && host_status, "read_status: NULL argument");
  SKGS_CHECK_HIP(hipMemcpyAsync(host_status, buf->geom, sizeof(skgs_status), hipMemcpyDeviceToHost, (hipStream_t) stream));
  return 0;
}

static int check_deform(const skgs_deform_inputs* in);

int skgs_rasterize_backward(const skgs_raster_inputs* in, const skgs_raster_buffers* buf, const int32_t* radii,
    const float* out_opacity, const skgs_raster_grads* gr, skgs_stream_t stream) {
  if (check_inputs(in) || check_buffers(in, buf, true)) return 1;
  SKGS_REQUIRE(gr != nullptr, "grads struct is NULL");
  if (in->P == 0) {  // (the deform backward job's bone gradients are always written completely)
    if (const skgs_deform_backward_job* dj0 = gr->deform_backward_job) {
      SKGS_REQUIRE(dj0->in && dj0->g_bone_T && dj0->g_bone_drot && dj0->g_bone_dscale, "deform_backward_job: NULL argument");
      hipStream_t s0 = (hipStream_t) stream;
      if (fill_u32(dj0->g_bone_T, 0u, (size_t) dj0->in->M * 7, s0) || fill_u32(dj0->g_bone_drot, 0u, (size_t) dj0->in->M * 4, s0) ||
          fill_u32(dj0->g_bone_dscale, 0u, (size_t) dj0->in->M * 3, s0))
        return 1;
    }
    if (const skgs_sp_skinning_job* sj0 = gr->sp_skinning_job)  // (empty lists: the superpoint gradients come out zero)
      return sp_skinning_check(*sj0) || launch_sp_skinning_rest(*sj0, (hipStream_t) stream);
    return 0;
  }
  SKGS_REQUIRE(radii && out_opacity, "radii / out_opacity are required");
  SKGS_REQUIRE(gr->dL_dout_color, "dL_dout_color is required");
  // (with a job attached the per-Gaussian gradients it consumes in registers need no array: any of them may be NULL)
  SKGS_REQUIRE(gr->dL_dmeans2D && ((gr->deform_backward_job || gr->sp_skinning_job) ||
                                      (gr->dL_dcolors && gr->dL_dopacity && gr->dL_dmeans3D && gr->dL_dcov3D && gr->dL_dscales &&
                                          gr->dL_drotations)),
      "gradient outputs are required");
  SKGS_REQUIRE(!(in->sh && in->sh_coeffs > 0) || gr->dL_dsh || gr->dL_dsh_factors,
      "dL_dsh (or dL_dsh_factors) is required when sh is given");
  SKGS_REQUIRE(gr->dL_dsh_factors || (in->sh_rest != nullptr) == (gr->dL_dsh_rest != nullptr),
      "dL_dsh_rest goes with sh_rest (split SH storage)");
  SKGS_REQUIRE(!gr->dL_dsh_factors || (!gr->dL_dsh && !gr->dL_dsh_rest), "dL_dsh_factors replaces dL_dsh / dL_dsh_rest");
  SKGS_REQUIRE(gr->workspace && gr->workspace_bytes >= skgs_backward_workspace_bytes(in->P), "workspace too small");
  hipStream_t s = (hipStream_t) stream;
  GeomView g    = geom_view(buf->geom);
  ImgView im    = img_view(buf->img, in->image_width, in->image_height);
  BinView b     = bin_view(buf->binning, buf->binning_bytes);
  const skgs_deform_backward_job* dj = gr->deform_backward_job;
  if (dj) {  // the skinning backward rides on the per-Gaussian launch (same checks as skgs_lbs_deform_backward_logits)
    SKGS_REQUIRE(dj->in != nullptr, "deform_backward_job: in is NULL");
    if (check_deform(dj->in)) return 1;
    SKGS_REQUIRE(dj->in->P == in->P && dj->in->live_count == in->live_count, "deform_backward_job: P / live_count differ from the rasterizer's");
    SKGS_REQUIRE(!dj->in->largest, "deform_backward_job: warp_method `largest` is served by the sp_skinning_job only");
    SKGS_REQUIRE(dj->in->M <= deform_backward_job_max_bones() && dj->in->K <= deform_backward_job_max_k(),
        "deform_backward_job: needs M <= %d, K <= %d (got %d, %d)", deform_backward_job_max_bones(), deform_backward_job_max_k(),
        dj->in->M, dj->in->K);
    SKGS_REQUIRE(dj->g_bone_T && dj->g_bone_drot && dj->g_bone_dscale, "deform_backward_job: bone gradient outputs are required");
    SKGS_REQUIRE(dj->g_xyz && dj->g_log_scale && dj->g_rot && dj->g_opacity_logit, "deform_backward_job: gradient outputs are required");
    SKGS_REQUIRE(dj->g_sp_W || dj->g_logits, "deform_backward_job: one of g_sp_W / g_logits is required");
    SKGS_REQUIRE(dj->workspace && dj->workspace_bytes >= deform_backward_workspace_bytes(in->P, dj->in->M),
        "deform_backward_job: workspace too small (skgs_lbs_deform_backward_workspace_bytes)");
    SKGS_REQUIRE(in->scales && in->rotations, "deform_backward_job: not with cov3D_precomp");
  }
  const skgs_sp_skinning_job* sj = gr->sp_skinning_job;
  if (sj) {  // the superpoint stage's rows pass rides on the per-Gaussian launch (checks of skgs_sp_skinning_backward)
    SKGS_REQUIRE(!dj, "sp_skinning_job: not together with deform_backward_job");
    if (sp_skinning_check(*sj)) return 1;
    SKGS_REQUIRE(sj->in->P == in->P && in->live_count == nullptr, "sp_skinning_job: P differs from the rasterizer's / no row capacity");
    SKGS_REQUIRE(in->scales && in->rotations, "sp_skinning_job: not with cov3D_precomp");
  }
  if (!gr->workspace_is_zero && fill_u32(gr->workspace, 0u, (size_t) in->P * GRAD_ROW, s)) return 1;
  if (launch_render_backward(*in, g, im, b, out_opacity, gr->dL_dout_color, gr->dL_dout_opacity, gr->dL_dout_extra,
          gr->workspace, s))
    return 1;
  if (launch_preprocess_backward(*in, g, radii, *gr, s)) return 1;
  if (sj) return launch_sp_skinning_rest(*sj, s);
  return dj ? launch_deform_backward_finalize(*dj->in, dj->workspace, dj->g_bone_T, dj->g_bone_drot, dj->g_bone_dscale, s) : 0;
}

int skgs_rasterize_extra_forward(int32_t W, int32_t H, int32_t P, int32_t E, const float* extra,
    const skgs_raster_buffers* buf, float* pixel_extra, skgs_stream_t stream) {
  SKGS_REQUIRE(buf && buf->geom && buf->img, "buffers are required");
  SKGS_REQUIRE(E > 0 && extra && pixel_extra, "Error shape for extras");
  if (P == 0) return 0;
  GeomView g = geom_view(buf->geom);
  ImgView im = img_view(buf->img, W, H);
  BinView b  = bin_view(buf->binning, buf->binning_bytes);
  return launch_extra_forward(W, H, P, E, extra, g, im, b, pixel_extra, (hipStream_t) stream);
}

int skgs_rasterize_extra_backward(int32_t W, int32_t H, int32_t P, int32_t E, const float* extra, const float* out_opacity,
    const float* grad_pixel_extra, const skgs_raster_buffers* buf, float* grad_means2D, float* grad_conic,
    float* grad_opacity, float* dL_dextra, skgs_stream_t stream) {
  SKGS_REQUIRE(buf && buf->geom && buf->img, "buffers are required");
  SKGS_REQUIRE(E > 0 && extra && grad_pixel_extra && out_opacity, "Error shape for extras");
  SKGS_REQUIRE(grad_means2D && grad_conic && grad_opacity && dL_dextra, "gradient outputs are required");
  if (P == 0) return 0;
  hipStream_t s = (hipStream_t) stream;
  GeomView g = geom_view(buf->geom);
  ImgView im = img_view(buf->img, W, H);
  BinView b  = bin_view(buf->binning, buf->binning_bytes);
  if (fill_u32(dL_dextra, 0u, (size_t) P * E, s)) return 1;
  return launch_extra_backward(W, H, P, E, extra, out_opacity, grad_pixel_extra, g, im, b, grad_means2D, grad_conic,
      grad_opacity, dL_dextra, s);
}

int skgs_topk_weights(int32_t topk, int32_t W, int32_t H, int32_t P, const skgs_raster_buffers* buf, int32_t* top_indices,
    float* top_weights, skgs_stream_t stream) {
  SKGS_REQUIRE(buf && buf->geom && buf->img, "buffers are required");
  SKGS_REQUIRE(topk > 0 && top_indices && top_weights, "topk outputs are required");
  hipStream_t s = (hipStream_t) stream;
  if (P == 0) {
    if (fill_u32(top_indices, 0xffffffffu, (size_t) W * H * topk, s)) return 1;
    if (fill_u32(top_weights, 0u, (size_t) W * H * topk, s)) return 1;
    return 0;
  }
  GeomView g = geom_view(buf->geom);
  ImgView im = img_view(buf->img, W, H);
  BinView b  = bin_view(buf->binning, buf->binning_bytes);
  return launch_topk(topk, W, H, g, im, b, top_indices, top_weights, s);
}

int skgs_render_census(int32_t W, int32_t H, const skgs_raster_buffers* buf, float* out_color, float* out_opacity,
    uint32_t* census, skgs_stream_t stream) {
  SKGS_REQUIRE(buf && buf->geom && buf->img && buf->binning, "buffers of a finished forward are required");
  SKGS_REQUIRE(out_color && out_opacity && census, "census outputs are required");
  GeomView g = geom_view(buf->geom);
  ImgView im = img_view(buf->img, W, H);
  BinView b  = bin_view(buf->binning, buf->binning_bytes);
  return launch_render_census(W, H, g, im, b, out_color, out_opacity, census, (hipStream_t) stream);
}

int skgs_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, int32_t colmap, uint8_t* present,
    skgs_stream_t stream) {
  SKGS_REQUIRE(P == 0 || (means3D && viewmatrix && present), "mark_visible: NULL argument");
  return launch_mark_visible(P, means3D, viewmatrix, colmap, present, (hipStream_t) stream);
}

static int check_deform(const skgs_deform_inputs* in) {
  SKGS_REQUIRE(in != nullptr, "deform inputs struct is NULL");
  SKGS_REQUIRE(in->P >= 0 && in->K >= 1 && in->M >= 1, "deform: need P >= 0, K >= 1, M >= 1");
  if (in->P == 0) return 0;
  SKGS_REQUIRE(in->points && in->weights && in->indices && in->bone_T && in->bone_drot && in->bone_dscale,
      "deform: points / weights / indices / bone tensors are required");
  SKGS_REQUIRE(in->log_scale && in->rot && in->opacity_logit, "deform: Gaussian parameter tensors are required");
  return 0;
}

int skgs_lbs_deform_forward(const skgs_deform_inputs* in, float* means, float* scales, float* rotations, float* opacity,
    float* d_xyz, float* d_rot, float* d_scale, skgs_stream_t stream) {
  if (check_deform(in)) return 1;
  SKGS_REQUIRE(in->P == 0 || (in->xyz && means && scales && rotations && opacity), "deform: outputs are required");
  return launch_deform_forward(*in, means, scales, rotations, opacity, d_xyz, d_rot, d_scale, (hipStream_t) stream);
}

size_t skgs_lbs_deform_backward_workspace_bytes(int32_t P, int32_t M) {
  return deform_backward_workspace_bytes(P < 0 ? 0 : P, M < 0 ? 0 : M);
}

int skgs_lbs_deform_backward(const skgs_deform_inputs* in, const float* g_means, const float* g_scales,
    const float* g_rotations, const float* g_opacity, float* g_weights, float* g_bone_T, float* g_bone_drot,
    float* g_bone_dscale, float* g_xyz, float* g_log_scale, float* g_rot, float* g_opacity_logit, void* workspace,
    size_t workspace_bytes, skgs_stream_t stream) {
  if (check_deform(in)) return 1;
  SKGS_REQUIRE(!in->largest, "deform backward: warp_method `largest` is served by skgs_sp_skinning_backward / sp_skinning_job only");
  SKGS_REQUIRE(in->P == 0 || (g_means && g_scales && g_rotations && g_opacity), "deform: upstream gradients are required");
  SKGS_REQUIRE(g_bone_T && g_bone_drot && g_bone_dscale, "deform: bone gradient outputs are required");
  SKGS_REQUIRE(in->P == 0 || (g_weights && g_xyz && g_log_scale && g_rot && g_opacity_logit),
      "deform: gradient outputs are required");
  SKGS_REQUIRE(in->P == 0 || (workspace && workspace_bytes >= deform_backward_workspace_bytes(in->P, in->M)),
      "deform backward: workspace too small (skgs_lbs_deform_backward_workspace_bytes)");
  return launch_deform_backward(*in, g_means, g_scales, g_rotations, g_opacity, g_weights, g_bone_T, g_bone_drot,
      g_bone_dscale, g_xyz, g_log_scale, g_rot, g_opacity_logit, workspace, (hipStream_t) stream);
}

int skgs_lbs_deform_backward_logits(const skgs_deform_inputs* in, const float* g_means, const float* g_scales,
    const float* g_rotations, const float* g_opacity, float* g_weights, float* g_bone_T, float* g_bone_drot,
    float* g_bone_dscale, float* g_xyz, float* g_log_scale, float* g_rot, float* g_opacity_logit, float* g_sp_W,
    float* g_logits, void* workspace, size_t workspace_bytes, skgs_stream_t stream) {
  if (check_deform(in)) return 1;
  SKGS_REQUIRE(!in->largest, "deform backward: warp_method `largest` is served by skgs_sp_skinning_backward / sp_skinning_job only");
  SKGS_REQUIRE(in->P == 0 || (g_means && g_scales && g_rotations && g_opacity), "deform: upstream gradients are required");
  SKGS_REQUIRE(g_bone_T && g_bone_drot && g_bone_dscale, "deform: bone gradient outputs are required");
  SKGS_REQUIRE(in->P == 0 || (g_xyz && g_log_scale && g_rot && g_opacity_logit), "deform: gradient outputs are required");
  SKGS_REQUIRE(g_sp_W || g_logits, "deform backward (logits): one of g_sp_W / g_logits is required");
  SKGS_REQUIRE(in->P == 0 || (workspace && workspace_bytes >= deform_backward_workspace_bytes(in->P, in->M)),
      "deform backward: workspace too small (skgs_lbs_deform_backward_workspace_bytes)");
  return launch_deform_backward(*in, g_means, g_scales, g_rotations, g_opacity, g_weights, g_bone_T, g_bone_drot,
      g_bone_dscale, g_xyz, g_log_scale, g_rot, g_opacity_logit, workspace, (hipStream_t) stream, g_sp_W, g_logits);
}

int skgs_sh_grad_from_factors(int32_t P, int32_t n_views, int32_t sh_degree, int32_t sh_coeffs, const float* factors,
    float* dL_dsh, float* dL_dsh_rest, skgs_stream_t stream) {
  SKGS_REQUIRE(P >= 0 && n_views >= 1, "sh_grad_from_factors: bad sizes");
  SKGS_REQUIRE(P == 0 || (factors && dL_dsh), "sh_grad_from_factors: NULL argument");
  return launch_sh_grad_from_factors(P, n_views, sh_degree, sh_coeffs, factors, dL_dsh, dL_dsh_rest, (hipStream_t) stream);
}

int skgs_knn_bones(int32_t P, int32_t M, int32_t K, int32_t dim, const float* points, const float* joints,
    float* out_dist, int64_t* out_idx, skgs_stream_t stream) {
  SKGS_REQUIRE(P == 0 || (points && joints && out_dist && out_idx), "knn_bones: NULL argument");
  SKGS_REQUIRE(M >= 1 && dim >= 1, "knn_bones: M and dim must be >= 1");
  return launch_knn_bones(P, M, K, dim, points, joints, out_dist, out_idx, (hipStream_t) stream);
}

int skgs_knn_dist_weights_forward(int32_t P, int32_t M, int32_t K, int32_t dim, const float* points, const float* joints,
    const float* kernel_radius, const float* kernel_weight, float temperature, int32_t raw_parameters, int64_t* out_idx,
    float* out_weights, float* out_dist, skgs_stream_t stream) {
  SKGS_REQUIRE(P == 0 || (points && joints && out_idx && out_weights && out_dist), "knn_dist_weights_forward: NULL argument");
  SKGS_REQUIRE(M >= 1, "knn_dist_weights_forward: M must be >= 1");
  SKGS_REQUIRE(kernel_radius || !kernel_weight, "knn_dist_weights_forward: kernel_weight needs kernel_radius");
  SKGS_REQUIRE(kernel_radius || temperature != 0.f, "knn_dist_weights_forward: the dist method needs a temperature != 0");
  return launch_knn_dist_weights_forward(P, M, K, dim, points, joints, kernel_radius, kernel_weight, temperature,
      raw_parameters ? 1 : 0, out_idx, out_weights, out_dist, (hipStream_t) stream);
}
size_t skgs_knn_dist_weights_workspace_bytes(int32_t P, int32_t M, int32_t dim) {
  return knn_dist_weights_workspace_bytes(P, M, dim);
}
int skgs_knn_dist_weights_backward(int32_t P, int32_t M, int32_t K, int32_t dim, const float* points, const float* joints,
    const float* kernel_radius, const float* kernel_weight, float temperature, int32_t raw_parameters,
    int32_t accumulate_joints, const float* weights, const int64_t* indices, const float* nn_dist, const float* g_weights,
    float* g_points, float* g_joints, float* g_kernel_radius, float* g_kernel_weight, void* workspace, size_t workspace_bytes,
    skgs_stream_t stream) {
  SKGS_REQUIRE(points && joints && (P == 0 || (weights && indices && nn_dist && g_weights)), "knn_dist_weights_backward: NULL argument");
  SKGS_REQUIRE(M >= 1, "knn_dist_weights_backward: M must be >= 1");
  SKGS_REQUIRE(kernel_radius || !kernel_weight, "knn_dist_weights_backward: kernel_weight needs kernel_radius");
  return launch_knn_dist_weights_backward(P, M, K, dim, points, joints, kernel_radius, kernel_weight, temperature,
      raw_parameters ? 1 : 0, accumulate_joints ? 1 : 0, weights, indices, nn_dist, g_weights, g_points, g_joints,
      g_kernel_radius, g_kernel_weight, workspace, workspace_bytes, (hipStream_t) stream);
}

int skgs_lbs_weights_backward_compact(int32_t P, int32_t K, const float* weights, const float* g_weights, float* g_logits,
    skgs_stream_t stream) {
  SKGS_REQUIRE(P == 0 || (weights && g_weights && g_logits), "lbs_weights_backward_compact: NULL argument");
  SKGS_REQUIRE(K >= 1, "lbs_weights_backward_compact: K must be >= 1");
  return launch_lbs_weights_backward_compact(P, K, weights, g_weights, g_logits, (hipStream_t) stream);
}

int skgs_lbs_logits_scatter(int32_t P, int32_t M, int32_t K, const int64_t* indices, const float* g_logits, float* g_sp_W,
    skgs_stream_t stream) {
  SKGS_REQUIRE(P == 0 || (indices && g_logits && g_sp_W), "lbs_logits_scatter: NULL argument");
  SKGS_REQUIRE(M >= 1 && K >= 1, "lbs_logits_scatter: M and K must be >= 1");
  return launch_lbs_logits_scatter(P, M, K, indices, g_logits, g_sp_W, (hipStream_t) stream);
}

int skgs_knn_lbs_weights(int32_t P, int32_t M, int32_t K, const float* points, const float* joints, const float* sp_W,
    int64_t* out_idx, float* out_weights, skgs_stream_t stream) {
  SKGS_REQUIRE(P == 0 || (points && joints && sp_W && out_idx && out_weights), "knn_lbs_weights: NULL argument");
  SKGS_REQUIRE(M >= 1, "knn_lbs_weights: M must be >= 1");
  return launch_knn_lbs_weights(P, M, K, points, joints, sp_W, out_idx, out_weights, (hipStream_t) stream);
}

int skgs_knn_lbs_deform_forward(int32_t P, int32_t M, int32_t K, const float* points, const float* joints, const float* sp_W,
    const float* bone_T, const float* bone_drot, const float* bone_dscale, const float* xyz, const float* log_scale,
    const float* rot, const float* opacity_logit, int64_t* out_idx, float* out_weights, float* means, float* scales,
    float* rotations, float* opacity, const int32_t* live_count, skgs_stream_t stream) {
  SKGS_REQUIRE(P == 0 || (points && joints && sp_W && bone_T && bone_drot && bone_dscale && xyz && log_scale && rot &&
                   opacity_logit && out_idx && out_weights && means && scales && rotations && opacity),
      "knn_lbs_deform_forward: NULL argument");
  SKGS_REQUIRE(M >= 1, "knn_lbs_deform_forward: M must be >= 1");
  return launch_knn_deform_forward(P, M, K, points, joints, sp_W, bone_T, bone_drot, bone_dscale, xyz, log_scale, rot,
      opacity_logit, out_idx, out_weights, means, scales, rotations, opacity, live_count, (hipStream_t) stream);
}

int skgs_lbs_weights_forward(int32_t P, int32_t M, int32_t K, const float* sp_W, const int64_t* indices, float* weights,
    skgs_stream_t stream) {
  SKGS_REQUIRE(P == 0 || (sp_W && indices && weights), "lbs_weights_forward: NULL argument");
  SKGS_REQUIRE(M >= 1, "lbs_weights_forward: M must be >= 1");
  return launch_lbs_weights_forward(P, M, K, sp_W, indices, weights, (hipStream_t) stream);
}

int skgs_lbs_weights_backward(int32_t P, int32_t M, int32_t K, const float* weights, const int64_t* indices,
    const float* g_weights, float* g_sp_W, skgs_stream_t stream) {
  SKGS_REQUIRE(P == 0 || (weights && indices && g_weights && g_sp_W), "lbs_weights_backward: NULL argument");
  SKGS_REQUIRE(M >= 1 && K >= 1, "lbs_weights_backward: M and K must be >= 1");
  return launch_lbs_weights_backward(P, M, K, weights, indices, g_weights, g_sp_W, (hipStream_t) stream);
}

}  // extern "C"
