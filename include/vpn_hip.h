/* vpn_hip.h — C ABI of libvpn_hip.so, the MI355X (gfx950) implementation of the
 * volumetric-primitive hot path (surface sampler, Chamfer nearest-neighbour
 * reduction, primitive soft raster), forward and analytic backward.
 *
 * Conventions:
 *  - every pointer is DEVICE memory owned by the caller (outputs and workspaces
 *    included); the library allocates no device memory and keeps no state that a
 *    RESULT depends on (contrast: the reference's module-global renderer,
 *    vertex_renderer.py:7,18).  What it does keep per process: the optional launch
 *    profiler's event list (vpn_profile_enable/read), two "dynamic LDS limit
 *    already raised" marks and the cached VPN_CHAMFER_MODE / VPN_CHAMFER_R
 *    environment overrides -- one process per GPU is the intended deployment,
 *    several host threads driving one library instance are not supported;
 *  - all tensors are contiguous fp32 unless stated, indices are int32;
 *  - `stream` is a hipStream_t passed as void*; kernels are only enqueued, no
 *    entry point synchronises with the host (the reference syncs through
 *    .item() at vertex_renderer.py:30-35 and cuboid.py:96);
 *  - return value: 0 on success, a positive hipError_t value if a launch failed,
 *    or a negative VPN_E_* code for rejected arguments.  Unlike the reference's
 *    emd extension (emd_cuda.cu:236-281, result ignored at emd_module.py:56) the
 *    host binding must raise on non-zero.
 *
 * Each entry point cites the reference interface it replaces (file:line relative
 * to the reference root).  The Python binding a maintainer adds is in
 * INTEGRATION.md (ctypes).
 */
#ifndef VPN_HIP_H
#define VPN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 5 (round 4): vpn_vpdiv_fwd, vpn_camera_matrix, vpn_trainstep_finalize, vpn_trainstep_bwd (the reference's whole training
 * step in one autograd node).  4 (round 3): raster records are 16 float4 per primitive (vpn_raster_records_size grew), tile_order is a buffer of
 * 48-byte tile entries (vpn_raster_order_size, K <= 64).  3: vpn_raster_total_fwd_fin, vpn_hotpath_chamfer_fwd, the mesh
 * entry points. */
#define VPN_ABI_VERSION 5

/* primitive kinds (reference: train.py:106-116 cuboids first, then spheres, cones are stubs) */
#define VPN_SPHERE 0
#define VPN_CUBOID 1

/* negative return codes */
#define VPN_E_BADARG  (-1)   /* null pointer / non-positive size              */
#define VPN_E_TOOBIG  (-2)   /* size above a documented limit                 */

/* packed primitive parameters: [B, K, 10] = (v0 v1 v2 | q0 q1 q2 q3 | t0 t1 t2)
 * — the reference passes K separate (B,3),(B,4),(B,3) tensors (train.py:117).  */
#define VPN_PARAM_STRIDE 10
/* largest K the raster / sampler stage in LDS */
#define VPN_MAX_PRIMS 1024

int vpn_abi_version(void);
/* static description of a code returned by any entry point below */
const char* vpn_error_string(int code);

/* Optional per-kernel timing for benchmarks: while enabled, every kernel the library launches is
 * bracketed by a pair of HIP events on its stream (do not enable under graph capture).
 * vpn_profile_enable(on) clears the records; vpn_profile_read synchronises and returns, per kernel
 * name (newline separated in `names`), the mean duration in ms and the number of launches. */
int vpn_profile_enable(int on);
int vpn_profile_read(char* names, int names_len, float* mean_ms, int* calls, int max_entries);

/* ------------------------------------------------------------------ sampler
 * Replaces Sampling.sphere_sampling / cuboid_sampling (modules/sampling/
 * sampling.py:11-37, sphere.py:22-43, cuboid.py:8-101), transform_points
 * (modules/transform/transform.py:6-9) and the per-primitive loop + torch.cat of
 * sample_predict_points (train.py:105-120) in one launch.
 *   params [B,K,10], kinds [K] int32 (device), points [B, K*n, 3] primitive-major.
 *   u: explicit uniform draws [B,K,n,3] (sphere: u0 = elev draw, u1 = azim draw,
 *      sphere.py:26-27; cuboid: the three draws of cuboid.py:66), or NULL to
 *      generate them in-kernel with Philox4x32-10 keyed by (seed, sample_base+b,
 *      k, point) so the result does not depend on how the batch is sharded.
 *   seed_dev: NULL, or a DEVICE uint64 read when the kernel runs and added to `seed`: a step counter the
 *      caller bumps on the stream, so that a captured HIP graph replayed N times draws N different point
 *      sets like the reference's per-step torch.rand (sphere.py:26-27) instead of freezing the seed at capture.
 */
int vpn_sample_fwd(const float* params, const int32_t* kinds, const float* u,
                   uint64_t seed, const uint64_t* seed_dev, uint64_t sample_base, int B, int K, int n,
                   float* points, void* stream);
/* grad_params [B,K,10] is WRITTEN (not accumulated).  (seed, *seed_dev) must be what the forward saw. */
int vpn_sample_bwd(const float* params, const int32_t* kinds, const float* u,
                   uint64_t seed, const uint64_t* seed_dev, uint64_t sample_base, int B, int K, int n,
                   const float* grad_points, float* grad_params, void* stream);
/* Backward of ChamferDistanceLoss(sample(params), gt_points) straight to grad_params [B,K,10], for the pair of
 * calls train.py:117-120 + :160-161 makes: equals vpn_chamfer_bwd (gradient of the predicted cloud only, GT gets
 * none) followed by vpn_sample_bwd, without the [B,K*n,3] point gradient in between and in a fixed summation
 * order.  points [B,K*n,3] = what vpn_sample_fwd produced from the same (params, kinds, u | seed, sample_base);
 * dist/idx from vpn_chamfer_fwd*(points, gt_points); grad_loss_b [B] as in vpn_chamfer_bwd.
 * M <= 7680 (VPN_E_TOOBIG beyond: use the two separate calls). */
int vpn_sample_chamfer_bwd(const float* params, const int32_t* kinds, const float* u,
                           uint64_t seed, const uint64_t* seed_dev, uint64_t sample_base, int B, int K, int n,
                           const float* points, const float* gt_points, int M,
                           const float* dist1, const int32_t* idx1, const float* dist2, const int32_t* idx2,
                           const float* grad_loss_b, float w1, float w2, float* grad_params, void* stream);

/* ---------------------------------------------------------------- transform
 * Replaces transform_points / rotate_points (modules/transform/transform.py:6-9,
 * rotate.py:7-25, translate.py:4-8): out = R(q) p + t.  t may be NULL (pure
 * rotation).  points/out [B,N,3], q [B,4], t [B,3].
 */
int vpn_transform_fwd(const float* points, const float* q, const float* t,
                      int B, int N, float* out, void* stream);
/* grad_points [B,N,3], grad_q [B,4], grad_t [B,3] are written; any may be NULL. */
int vpn_transform_bwd(const float* points, const float* q, const float* grad_out,
                      int B, int N, float* grad_points, float* grad_q, float* grad_t,
                      void* stream);

/* ------------------------------------------------------------------ Chamfer
 * Replaces the dense B*N*M expression of ChamferDistanceLoss.forward
 * (modules/loss/chamfer_distance.py:14-23).  p1 [B,N,3], p2 [B,M,3].
 *   dist1[b,i] = min_j ||p1_i - p2_j|| (NON-squared), idx1 = argmin_j with the
 *   reference's tie rule (lowest index among equal sqrt values), dist2/idx2 the
 *   other direction.  Bit-exact with the fp32 reference expression.
 */
int vpn_chamfer_fwd(const float* p1, const float* p2, int B, int N, int M,
                    float* dist1, int32_t* idx1, float* dist2, int32_t* idx2, void* stream);
/* One direction only (one kernel launch): for every query point the nearest target point.
 * queries [B,Nq,3], targets [B,Nt,3] -> dist [B,Nq], idx [B,Nq].  vpn_chamfer_fwd is two of these. */
int vpn_chamfer_nn(const float* queries, const float* targets, int B, int Nq, int Nt,
                   float* dist, int32_t* idx, void* stream);
/* Both directions with a caller-provided workspace of vpn_chamfer_workspace(B,N,M) bytes: each
 * scan strategy is selectable and every strategy returns the same bits as vpn_chamfer_fwd:
 *   mode 1  brute force (workspace unused, may be NULL);
 *   mode 2  box-pruned: clouds Morton-sorted per call, target chunks farther than the current best skipped;
 *   mode 3  matrix-pipe filter: bf16 MFMA (fp32 coordinates split exactly into three bf16 pieces) evaluates
 *           |b|^2 - 2a.b for 32x32 pairs, the candidates are re-evaluated with the exact separately-rounded
 *           d2 (a rigorous error band decides when a second block or a full exact rescan is needed);
 *   mode 4  the same filter with fp32-input MFMA;
 *   mode 5  mode 3 over Morton-sorted clouds with per-block boxes (target blocks that cannot hold a nearer point skipped);
 *   mode 6  the filter with ONE fp16 MFMA per 32x32 block: coordinates scaled by 2^11 and split into two fp16
 *           pieces (32-byte rows); queries or clouds with |p|^2 > 64 are outside its domain and are finished by
 *           the exact rescan (results unchanged, only slower);
 *   mode 7  mode 6 with the features of both clouds already in the workspace (written by vpn_hotpath_sample_fwd
 *           for p1 = its points and p2 = its gt_points, earlier on the same stream);
 *   mode 0  automatic (mode 6 for large clouds when a workspace is given, else mode 1). */
size_t vpn_chamfer_workspace(int B, int N, int M);
/* workspace_bytes: size of `workspace`; VPN_E_BADARG if a mode that uses it is given fewer than
 * vpn_chamfer_workspace(B,N,M) bytes or a pointer that is not 16-byte aligned (the filtered scans fetch row
 * tiles without bounds checks inside it). */
int vpn_chamfer_fwd_ws(const float* p1, const float* p2, int B, int N, int M,
                       float* dist1, int32_t* idx1, float* dist2, int32_t* idx2,
                       void* workspace, size_t workspace_bytes, int mode, void* stream);
/* loss_b[b] = w1*mean_i dist1[b,i] + w2*mean_j dist2[b,j]   (chamfer_distance.py:25-28) */
int vpn_chamfer_loss(const float* dist1, const float* dist2, int B, int N, int M,
                     float w1, float w2, float* loss_b, void* stream);
/* Backward of loss_b w.r.t. the points given grad_loss_b [B].  grad_p1 [B,N,3] /
 * grad_p2 [B,M,3] are written; either may be NULL.  A coincident pair gives NaN
 * like the reference's autograd (0/0). */
int vpn_chamfer_bwd(const float* p1, const float* p2,
                    const float* dist1, const int32_t* idx1,
                    const float* dist2, const int32_t* idx2,
                    const float* grad_loss_b, int B, int N, int M, float w1, float w2,
                    float* grad_p1, float* grad_p2, void* stream);

/* ------------------------------------------------------------------- raster
 * Replaces VertexRenderer.render (modules/render/vertex_renderer.py:14-26) and the
 * per-sample loop of SilhouetteLoss.forward (modules/loss/silhouette.py:16-20):
 * one launch renders the whole batch straight from the primitive parameters (the
 * reference meshes the primitives first, modules/meshing/sphere.py:8-27, and
 * rasterises the mesh with kaolin's DIBRenderer, which is not in its tree).
 *   cam [B,3] = (dist, elev_deg, azim_deg)   (vertex_renderer.py:18)
 *   alpha, depth [B,H,W]; aux [B,3,H,W] = (prod(1-a), zbar, sum w) saved for bwd;
 *   records: vpn_raster_records_size(B,K,H,W) bytes, 16-byte aligned, written by fwd (the
 *   per-primitive camera-space ray coefficients and culling conics, then the per-tile
 *   visibility masks) and read again by bwd — keep it alive with aux.
 */
size_t vpn_raster_records_size(int B, int K, int H, int W);
int vpn_raster_fwd(const float* params, const int32_t* kinds, const float* cam,
                   int B, int K, int H, int W, float sigma, float gamma, float z_far,
                   float* alpha, float* depth, float* aux, void* records, void* stream);
/* bytes of workspace vpn_raster_bwd needs (16-byte aligned, contents scratch) */
size_t vpn_raster_bwd_workspace(int B, int K, int H, int W);
/* grad_alpha / grad_depth [B,H,W] (either may be NULL = zero); grad_params
 * [B,K,10] is written.  Bitwise reproducible (no atomics). */
int vpn_raster_bwd(const float* params, const int32_t* kinds, const float* cam,
                   int B, int K, int H, int W, float sigma, float gamma, float z_far,
                   const float* aux, const void* records,
                   const float* grad_alpha, const float* grad_depth,
                   void* workspace, float* grad_params, void* stream);

/* ------------------------------------------------ raster with fused image losses
 * SilhouetteLoss.forward (modules/loss/silhouette.py:13-23: render, then L1Loss or MSELoss mean
 * against the GT silhouette) in one pass, optionally with an L1 depth loss: the losses are
 * evaluated where the pixel is produced, so alpha/depth and their gradients never travel through HBM.
 *   gt_sil, gt_depth [B,H,W] (either may be NULL -> that loss is 0); sil_mse: 0 = L1, 1 = MSE;
 *   losses [4] = (mean silhouette loss, mean |depth - gt_depth|, their sum, 0);
 *   loss_ws: vpn_raster_loss_workspace(B,H,W) bytes of scratch (16-byte aligned); aux/records as for vpn_raster_fwd.
 */
size_t vpn_raster_loss_workspace(int B, int H, int W);
int vpn_raster_loss_fwd(const float* params, const int32_t* kinds, const float* cam,
                        int B, int K, int H, int W, float sigma, float gamma, float z_far,
                        const float* gt_sil, const float* gt_depth, int sil_mse,
                        float* aux, void* records, void* loss_ws, float* losses, void* stream);
/* grad_losses [2] (device): upstream gradients of the two scalar losses; workspace as for
 * vpn_raster_bwd; grad_params [B,K,10] is written, or added to when accumulate != 0 (so the gradient of
 * the sampler + Chamfer branch and of the raster branch meet without an extra pass). */
int vpn_raster_loss_bwd(const float* params, const int32_t* kinds, const float* cam,
                        int B, int K, int H, int W, float sigma, float gamma, float z_far,
                        const float* aux, const void* records,
                        const float* gt_sil, const float* gt_depth, int sil_mse,
                        const float* grad_losses, void* workspace, float* grad_params, int accumulate,
                        void* stream);

/* ---- the training step's image losses, forward and backward in ONE pass (train.py:243-262, :176)
 *   total_img = w_sil * SilhouetteLoss(render(params), gt_sil) + w_dep * mean |depth - gt_depth|
 * vpn_raster_total_fwd renders, evaluates both losses and, in the same kernel, the gradient of total_img w.r.t. the
 * ray coefficients (for an upstream gradient of 1; the result is linear in it): per-tile loss sums go to loss_ws,
 * gradient partials to `workspace` (vpn_raster_bwd_workspace bytes); no aux tensor exists.
 * vpn_loss_finalize turns the per-tile sums (and, when dist1/dist2 [B,N]/[B,M] are given, the per-sample Chamfer
 * loss cd_b = cd_w1 mean_i dist1 + cd_w2 mean_j dist2, chamfer_distance.py:25-28) into
 *   losses[4] = (silhouette loss, depth loss, w_cd * mean_b cd_b + w_sil * [0] + w_dep * [1], mean_b cd_b)
 * in a fixed summation order; loss_b [B] (optional) receives cd_b.  H = W = 0: no image losses (Chamfer only; loss_ws
 * then needs 16 + 16 B bytes, zeroed once by the caller).
 * vpn_raster_total_bwd (when backward runs): grad_params (+)= (*grad_total) * d total_img / d params from the
 * partials; grad_total is a DEVICE scalar (NULL = 1). */
/* records_ready != 0: `records` were already written (and the counter at the head of loss_ws zeroed) by
 * vpn_hotpath_sample_fwd for the same (params, kinds, cam, H, W, sigma): the record launch is skipped. */
int vpn_raster_total_fwd(const float* params, const int32_t* kinds, const float* cam,
                         int B, int K, int H, int W, float sigma, float gamma, float z_far,
                         const float* gt_sil, const float* gt_depth, int sil_mse, float w_sil, float w_dep,
                         void* records, void* loss_ws, void* workspace, int records_ready, void* stream);
/* vpn_raster_total_fwd and vpn_loss_finalize in ONE launch (what SilhouetteLoss and the training step use): the tile
 * wave that completes a sample sums that sample's tile losses, the one that completes the batch writes losses[4] as
 * vpn_loss_finalize defines them (fixed summation order: bitwise reproducible).  chamfer_ws != NULL: the workspace a
 * preceding vpn_chamfer_fwd_ws call on this stream ran its fp16 filter in (vpn_hotpath_fused_features(B, K, n, M) == 1,
 * modes 0 / 6 / 7): the per-sample Chamfer term cd_b = cd_w1 mean_i dist1 + cd_w2 mean_j dist2 (chamfer_distance.py:25-28)
 * is taken from the per-workgroup sums that scan left there, N and M are its cloud sizes; NULL: image losses only
 * (cd = 0).  loss_b [B] (optional) receives cd_b.  seed_advance (optional): a device uint64 that is incremented once
 * when the batch is complete -- the step counter vpn_hotpath_sample_fwd was given as seed_dev, so the next step draws
 * fresh surface points without a kernel of its own; the seed this step used stays readable at (char*)loss_ws + 8.
 * Ordering the in-kernel finalisation rests on: a tile wave publishes its two loss sums with write-through (sc1) stores,
 * waits for their acknowledgement (vmcnt(0)) and only then adds to its sample's arrival counter with a relaxed
 * agent-scope add; the wave whose add completes the sample reads the sums back with sc1 loads.  That is the hand-off
 * form gfx950 executes correctly (MI355X_MICROARCH.md lists it as valid, "not an architectural guarantee"); it is NOT
 * the HIP memory model's release / acquire pairing, which a build with -DVPN_STRICT_ORDER uses instead (same results,
 * 6x the kernel time: an L2 write-back and an L1 invalidate per tile wave).  Gradients do not depend on it.
 * tile_order (optional, needs records_ready and K <= 64): the tile entries vpn_hotpath_chamfer_fwd wrote -- launch order
 * of the tile waves (heaviest tile first), tile masks and quadrant masks; the tile masks inside `records` are then taken
 * as written too. */
int vpn_raster_total_fwd_fin(const float* params, const int32_t* kinds, const float* cam,
                             int B, int K, int H, int W, float sigma, float gamma, float z_far,
                             const float* gt_sil, const float* gt_depth, int sil_mse, float w_sil, float w_dep,
                             void* records, void* loss_ws, void* workspace, int records_ready,
                             const void* chamfer_ws, size_t chamfer_ws_bytes, int N, int M, float cd_w1, float cd_w2,
                             float w_cd, float* losses, float* loss_b, uint64_t* seed_advance, const void* tile_order,
                             void* stream);
/* vpn_sample_fwd of the training step: the same launch also writes the raster records of the primitives it samples
 * (one workgroup per (sample, primitive) in both) and zeroes the arrival counter of loss_ws.
 * chamfer_ws != NULL (allowed when vpn_hotpath_fused_features(B, K, n, M) is 1): the launch also writes, into that
 * Chamfer workspace (vpn_chamfer_workspace(B, K*n, M) bytes), the matrix-pipe filter's features of the cloud it
 * samples and of gt_points [B,M,3]; the caller then runs vpn_chamfer_fwd_ws(points, gt_points, ..., mode 7) on the same
 * workspace and stream, which skips its feature kernel (the points would be written and immediately re-read).
 * chamfer_ws == NULL: gt_points / M / chamfer_ws_bytes are ignored.  With a Chamfer workspace the records are built
 * by workgroups of their own at the head of the launch (one lane per primitive), not by the sampler workgroups. */
int vpn_hotpath_sample_fwd(const float* params, const int32_t* kinds, const float* u,
                           uint64_t seed, const uint64_t* seed_dev, uint64_t sample_base, int B, int K, int n,
                           float* points, const float* cam, int H, int W, float sigma, void* records, void* loss_ws,
                           const float* gt_points, int M, void* chamfer_ws, size_t chamfer_ws_bytes,
                           void* stream);
int vpn_hotpath_fused_features(int B, int K, int n, int M);
/* The Chamfer scan of the training step (vpn_chamfer_fwd_ws with the fp16 filter: mode 0 where it resolves to it, 6 or 7)
 * with a RIDER in the same launch: behind the scan's workgroups, in the tail of the launch where slots idle, one
 * workgroup per image tests that image's 16x16 tiles and their 8x8 quadrants against its K primitives (the visibility
 * test of the tile kernels), writes the tile masks into `records` (as written by vpn_hotpath_sample_fwd earlier on this
 * stream) and tile_order (vpn_raster_order_size bytes, 16-byte aligned): one 48-byte entry per (image, launch rank) =
 * (tile, number of visible primitives, their mask, four quadrant masks), the tiles of every image sorted by visible
 * primitives, heaviest first -- what a tile wave of vpn_raster_total_fwd_fin needs to know, in one load (the second half
 * of the buffer is the rider's scratch: the same entries by tile).  tile_order == NULL: exactly vpn_chamfer_fwd_ws.
 * VPN_E_TOOBIG if K > 64 or the image has more than 16384 tiles (the caller then runs without it). */
size_t vpn_raster_order_size(int B, int H, int W);
int vpn_hotpath_chamfer_fwd(const float* p1, const float* p2, int B, int N, int M, float* dist1, int32_t* idx1,
                            float* dist2, int32_t* idx2, void* workspace, size_t workspace_bytes, int mode,
                            void* records, int K, int H, int W, void* tile_order, void* stream);
int vpn_loss_finalize(void* loss_ws, int B, int H, int W, const float* dist1, const float* dist2, int N, int M,
                      float cd_w1, float cd_w2, float w_cd, float w_sil, float w_dep, float* losses, float* loss_b,
                      void* stream);
int vpn_raster_total_bwd(const float* params, const float* cam, int B, int K, int H, int W,
                         const void* records, const void* workspace, const float* grad_total,
                         float* grad_params, int accumulate, void* stream);
/* Backward of the whole training step in ONE launch: vpn_sample_chamfer_bwd and vpn_raster_total_bwd together
 * (both are one workgroup per (sample, primitive)):
 *   grad_params = d(Chamfer term)/d params [as vpn_sample_chamfer_bwd] + (*grad_total) * d(total_img)/d params.
 * grad_loss_b may be NULL: then d total / d loss_b = *grad_total for every sample, and the caller folds the constant
 * factors of the batch mean into the weights (w1 = cd_w1 * w_cd / B, w2 = cd_w2 * w_cd / B): no kernel in between. */
int vpn_hotpath_bwd(const float* params, const int32_t* kinds, const float* u,
                    uint64_t seed, const uint64_t* seed_dev, uint64_t sample_base, int B, int K, int n,
                    const float* points, const float* gt_points, int M,
                    const float* dist1, const int32_t* idx1, const float* dist2, const int32_t* idx2,
                    const float* grad_loss_b, float w1, float w2,
                    const float* cam, int H, int W, const void* records, const void* workspace,
                    const float* grad_total, float* grad_params, void* stream);

/* ---- fused camera transforms (row f3): view_to_obj_points / obj_to_view_points
 * (modules/transform/transform.py:21-47, :50-73; called on train.py:158 every step).
 * points [B,N,3]; dists, elevs, azims, angles [B] (degrees, as the reference's callers pass them; angles may
 * be NULL when to_object == 0).  to_object != 0: out = dist * R(-z,-e) R(y',-a) R(x,-angle) p;
 * to_object == 0: out = R(y',a) R(-z,e) p / dist.  One launch instead of 3-4 rotate_points + a scale.
 * The camera is data (dataset.py:145-165): only the points receive a gradient. */
int vpn_camera_transform_fwd(const float* points, const float* dists, const float* elevs, const float* azims,
                             const float* angles, int B, int N, int to_object, float* out, void* stream);
int vpn_camera_transform_bwd(const float* grad_out, const float* dists, const float* elevs, const float* azims,
                             const float* angles, int B, int N, int to_object, float* grad_points, void* stream);

/* ---- primitive -> mesh vertices (row f2: modules/meshing/sphere.py:8-27, cuboid.py:8-26, meshing.py:27-46)
 * verts [B, Ptot, 3]: verts[b][offsets[k] + p] = R(q_bk) (tpl[p] * v_bk) + t_bk, tpl = tpl_sphere [Ps,3] or
 * tpl_cuboid [Pc,3] by kinds[k]; offsets [K+1] int32 (device): vertex offsets of the composed mesh, offsets[K] = Ptot.
 * The reference parses the OBJ template from disk per (sample, primitive); here the templates are device arrays the
 * caller keeps (a template of a kind that does not occur may be NULL). */
int vpn_mesh_fwd(const float* params, const int32_t* kinds, const int32_t* offsets,
                 const float* tpl_sphere, const float* tpl_cuboid, int B, int K, int Ptot, float* verts, void* stream);
int vpn_mesh_bwd(const float* params, const int32_t* kinds, const int32_t* offsets,
                 const float* tpl_sphere, const float* tpl_cuboid, int B, int K, int Ptot,
                 const float* grad_verts, float* grad_params, void* stream);

/* ---- the triangle-mesh path: meshes that carry NO primitives (train_sphere.py:53-76,128: the 386-vertex sphere of
 * 386.obj deformed in place, sampled by kaolin's TriangleMesh.sample and rendered by VertexRenderer.render /
 * SilhouetteLoss.forward through kaolin's DIBRenderer, vertex_renderer.py:20-24).  kaolin is absent: the arithmetic is
 * this repository's specification (oracle.vpn_oracle.mesh_raster / mesh_sample), parity unpinned.
 * verts [B,P,3] fp32 (B meshes of one topology), faces [F,3] int32 (vertex indices, clamped to [0,P) on the device),
 * cam [B,3] = (dist, elev deg, azim deg) as vpn_raster_fwd.
 * vpn_mesh_raster_fwd: alpha [B,H,W] = 1 - prod_f (1 - sigmoid(+-d2_f / sigma)), d2_f = squared NDC distance of the
 *   pixel centre to the nearest edge of face f, + inside / - outside; workspace = vpn_mesh_raster_workspace(B, P) bytes
 *   (projected vertices, kept for the backward call).
 * vpn_mesh_raster_bwd: grad_verts [B,P,3] = d (sum grad_alpha . alpha) / d verts (same workspace as the forward call).
 * vpn_mesh_sample_fwd: n area-weighted uniform surface points per mesh: points [B,n,3], face_idx [B,n], bary [B,n,3]
 *   (the barycentric weights, for the backward); u [B,n,3] explicit uniforms or NULL = Philox(seed; mesh_base + b,
 *   slot 0xFFFFFFFF, point); cdf [B,F] scratch (cumulative face areas).
 * vpn_mesh_sample_bwd: grad_verts [B,P,3] = sum_i bary_i * grad_points_i (the face choice is not differentiated). */
size_t vpn_mesh_raster_workspace(int B, int P);
int vpn_mesh_raster_fwd(const float* verts, const int32_t* faces, const float* cam, int B, int P, int F, int H, int W,
                        float sigma, void* workspace, float* alpha, void* stream);
int vpn_mesh_raster_bwd(const float* verts, const int32_t* faces, const float* cam, int B, int P, int F, int H, int W,
                        float sigma, void* workspace, const float* alpha, const float* grad_alpha, float* grad_verts,
                        void* stream);
int vpn_mesh_sample_fwd(const float* verts, const int32_t* faces, const float* u, uint64_t seed, uint64_t mesh_base,
                        int B, int P, int F, int n, float* cdf, float* points, int32_t* face_idx, float* bary, void* stream);
int vpn_mesh_sample_bwd(const int32_t* faces, const int32_t* face_idx, const float* bary, const float* grad_points,
                        int B, int P, int F, int n, float* grad_verts, void* stream);

/* ---- head post-processing into packed primitive parameters (row f4)
 * restrict_range + split + restrict_volumes of the reference's model (modules/network/vpnet_one_resnet.py:34-41,
 * :67-85): volumes [B,3K], rotates [B,4K], translates [B,3K] (raw head outputs) -> params [B,K,10].
 * is_sigmoid != 0 (config.py:25): v = (sigmoid(x) + 0.1) / restrict[j], q = sigmoid(x), t = tanh(x);
 * else v = clamp(x, clamp_min + 1e-8, clamp_max) / restrict[j], q, t = clamp(x, -1, 1) (config.py:22-23, :26). */
int vpn_head_pack_fwd(const float* volumes, const float* rotates, const float* translates, int B, int K,
                      int is_sigmoid, float clamp_min, float clamp_max, float restrict0, float restrict1,
                      float restrict2, float* params, void* stream);
/* grad_params [B,K,10] -> gradients of the raw head outputs (any of the three may be NULL). */
int vpn_head_pack_bwd(const float* volumes, const float* rotates, const float* translates, const float* grad_params,
                      int B, int K, int is_sigmoid, float clamp_min, float clamp_max, float restrict0,
                      float restrict1, float restrict2, float* grad_volumes, float* grad_rotates,
                      float* grad_translates, void* stream);

/* ---- the reference's whole training step (train.py:243-262) around the hot path: BASELINE config C5
 *   total = L_VIEW_CD * ChamferDistanceLoss(pred, view_center)            (train.py:160)   [vpn_hotpath_*]
 *         + L_CAN_CD  * ChamferDistanceLoss(view_to_obj(pred), canonical) (train.py:158-161)
 *         + L_SIL     * SilhouetteLoss(primitives, silhouettes)           (train.py:176)   [vpn_raster_total_fwd_fin]
 *         + L_VP_DIV  * VPDiverseLoss(translates, view_center)            (train.py:185, vp_diverse.py:12-18)
 *         + L_EMD     * sqrt(EMD dist).mean()                             (train.py:193-195) [vpn_emd_fwd]
 * vpn_vpdiv_fwd: the nearest neighbours of VPDiverseLoss -- centres = params[b,k,7:10] (the reference cats the K
 *   translations), both directions: dist1/idx1 [B,K] (centre -> nearest ground-truth point), dist2/idx2 [B,M]; same
 *   arithmetic and tie rule as vpn_chamfer_fwd (chamfer_distance.py:14-23).  dist1 = idx1 = NULL: the centres' direction stays
 *   as per-slice results in `workspace` for vpn_trainstep_finalize to merge.
 * vpn_camera_matrix: mat [B,9] row-major with view_to_obj_points(p) = mat p (to_object != 0; transform.py:21-47) or
 *   obj_to_view_points(p) = mat p (to_object == 0; :50-73), the scale by dist included.
 * vpn_trainstep_finalize: out [6] = (L_VIEW_CD*view_cd, L_CAN_CD*obj_cd, L_SIL*sil, L_VP_DIV*vp_div, L_EMD*emd, their sum)
 *   from hot_losses [4] as vpn_raster_total_fwd_fin leaves them (w_cd = L_VIEW_CD, w_sil = L_SIL, w_dep = 0), the
 *   auction's dist [B,N], the object-centred cloud's nearest-neighbour distances cn_dist1 [B,N] / cn_dist2 [B,Mc] and
 *   the VP-diversity distances dv_dist1 [B,K] / dv_dist2 [B,M]; any of the three groups may be NULL (term = 0).
 *   Fixed summation order.  dv_workspace != NULL (then dv_dist1 == NULL): the workspace a preceding vpn_vpdiv_fwd call
 *   with dist1 = idx1 = NULL left its per-slice results in; they are merged into dv_dist1_out / dv_idx1_out [B,K] by the
 *   same per-sample pass (one launch less in the training step).
 * vpn_trainstep_bwd: grad_params [B,K,10] = (*grad_total) * d total / d params in ONE launch: vpn_hotpath_bwd (w1 =
 *   cd_w1 * L_VIEW_CD / B, w2 = cd_w2 * L_VIEW_CD / B; records / workspace NULL when the silhouette term is off) plus
 *   - the EMD term through the assignment (emd_cuda.cu:284-300 and the sqrt / mean of train.py:195): emd_coef = L_EMD / (B N);
 *   - the VP-diversity term, straight into the translations: dv_c1 = L_VP_DIV * 0.5 / (K B), dv_c2 = L_VP_DIV * 1.0 / (M B);
 *   - the object-centred Chamfer term through cn_mat [B,9] (cn_points = the transformed cloud, cn_gt [B,cn_M,3],
 *     its four nearest-neighbour arrays): cn_c1 = L_CAN_CD * cd_w1 / (N B), cn_c2 = L_CAN_CD * cd_w2 / (cn_M B).
 *   A group whose first pointer is NULL is skipped (the reference's default L_CAN_CD = 0 multiplies that gradient by 0). */
size_t vpn_vpdiv_workspace(int B, int K);          /* scratch of vpn_vpdiv_fwd (8-byte aligned) */
int vpn_vpdiv_fwd(const float* params, const float* gt_points, int B, int K, int M, float* dist1, int32_t* idx1,
                  float* dist2, int32_t* idx2, void* workspace, void* stream);
int vpn_camera_matrix(const float* dists, const float* elevs, const float* azims, const float* angles, int B,
                      int to_object, float* mat, void* stream);
size_t vpn_trainstep_workspace(int B);             /* scratch of vpn_trainstep_finalize (per-sample sums) */
int vpn_trainstep_finalize(const float* hot_losses, const float* emd_dist, const float* cn_dist1, const float* cn_dist2,
                           const float* dv_dist1, const float* dv_dist2, int B, int N, int M, int Mc, int K,
                           float w_view, float w_can, float w_sil, float w_div, float w_emd, float cd_w1, float cd_w2,
                           void* workspace, const void* dv_workspace, float* dv_dist1_out, int32_t* dv_idx1_out,
                           float* out, void* stream);
int vpn_trainstep_bwd(const float* params, const int32_t* kinds, uint64_t seed, const uint64_t* seed_dev,
                      uint64_t sample_base, int B, int K, int n, const float* points, const float* gt_points, int M,
                      const float* dist1, const int32_t* idx1, const float* dist2, const int32_t* idx2,
                      float w1, float w2, const float* cam, int H, int W, const void* records,
                      const void* workspace, const float* grad_total,
                      const float* emd_dist, const int32_t* emd_assign, float emd_coef,
                      const float* dv_dist1, const int32_t* dv_idx1, const float* dv_dist2, const int32_t* dv_idx2,
                      float dv_c1, float dv_c2,
                      const float* cn_points, const float* cn_gt, const float* cn_mat, const float* cn_dist1,
                      const int32_t* cn_idx1, const float* cn_dist2, const int32_t* cn_idx2, float cn_c1, float cn_c2,
                      int cn_M, float* grad_params, void* stream);

/* ---- Earth Mover's Distance, auction approximation (row f1)
 * Replaces emd.forward / emd.backward of the reference's CUDA extension (modules/loss/emd/emd_cuda.cu:228-282,
 * :302-316, bound in emd_module.py:56, :69).  xyz1, xyz2 [B,n,3] (the reference requires equal sizes,
 * emd_module.py:36); dist [B,n] = squared distance of every xyz1 point to its assigned xyz2 point;
 * assignment [B,n] int32.  The nine scratch tensors the reference's caller allocates (emd_module.py:44-54)
 * become one workspace of vpn_emd_workspace(B, n) bytes.  iters >= 1, eps >= 0.  n need not be a multiple
 * of 1024 and B is not limited to 512 (emd_module.py:38-39). */
/* max_group: cap on the number of workgroups that cooperate on one sample (0 = automatic: as many as can be
 * resident together by hipOccupancyMaxActiveBlocksPerMultiprocessor x #CU, power of two <= 16; 1 = one workgroup per
 * sample, nothing shared between workgroups).  The workgroups of a sample wait for each other's bids: the whole grid must
 * be resident, which the occupancy bound guarantees for a plain launch on a GPU that runs nothing else at the time; the
 * wait is bounded (~0.5 s: dist = NaN, assignment = -1 for that sample rather than a hang).  Pass 1 when other streams or
 * processes share the GPU.  VPN_EMD_COOP_LAUNCH=1 in the environment launches cooperatively (the runtime then checks
 * residency itself; the call falls back to 1 if it refuses) -- not the default: a cooperative launch in a process that has
 * captured a HIP graph slows every later dispatch of that process by ~50 us.  Results do not depend on any of this.
 * For 128 <= n <= 2048 the auction is one workgroup per CU (1024 threads, <= 128 VGPRs, 131 KB of LDS at n = 2048,
 * G = 4 at B = 64): kernels of ANOTHER stream that need <= 32 KB of LDS and <= 4 waves per SIMD run beside it, which is
 * how TrainStepLossFunction uses it -- launch the auction first, so its workgroups are placed while the CUs are empty (a
 * workgroup that must wait for a slot only makes its partners spin, bounded as above).  Environment, performance only:
 * VPN_EMD_FLAT_WORK (default 4000: own bidders x targets per bid from which a round bids in the balanced form),
 * VPN_EMD_FLAT_MIN (default 16; 0 = team form only), VPN_EMD_TNUM / VPN_EMD_TMAX (team size of the team form). */
size_t vpn_emd_workspace(int B, int n);
int vpn_emd_fwd(const float* xyz1, const float* xyz2, int B, int n, float eps, int iters,
                float* dist, int32_t* assignment, void* workspace, int max_group, void* stream);
/* grad_xyz1 [B,n,3] = 2 grad_dist (xyz1 - xyz2[assignment]) is written; xyz2 receives no gradient
 * (emd_module.py:66-70 returns zeros for it). */
int vpn_emd_bwd(const float* xyz1, const float* xyz2, const float* grad_dist, const int32_t* assignment,
                int B, int n, float* grad_xyz1, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VPN_HIP_H */
