// C ABI of libedgegan_hip.so (see include/edgegan_hip.h).  Translates exceptions to error codes.
#include <algorithm>
#include <cstdlib>
#include <sstream>

#include "../../include/edgegan_hip.h"
#include "engine.h"

using namespace eg;

struct eg_ctx {
  Ctx c;
  std::deque<Param> tmp_params;  // per-op test scaffolding
};

static thread_local std::string g_create_error;

static Config to_config(const eg_config* k) {
  EG_CHECK(k, "null eg_config");
  if (k->struct_size != (unsigned)sizeof(eg_config))
    fail(3, "eg_config layout mismatch: struct_size %u, this library was built with sizeof(eg_config) = %u (call eg_default_config "
            "first, and rebuild the binding against include/edgegan_hip.h)", k->struct_size, (unsigned)sizeof(eg_config));
  Config c;
  c.B = k->batch_size;
  c.H = k->height;
  c.W = k->width;
  c.num_classes = k->num_classes;
  c.z_dim = k->z_dim;
  c.gf = k->gf_dim;
  c.df = k->df_dim;
  c.c_dim = k->c_dim;
  c.use_image_dis = k->use_image_discriminator != 0;
  c.image_dis = k->image_dis_size;
  c.use_edge_dis = k->use_edge_discriminator != 0;
  c.edge_dis = k->edge_dis_size;
  c.lambda_gp = k->lambda_gp;
  c.zl_weight = k->zl_weight;
  c.lr = k->learning_rate;
  c.joint_dweight = k->joint_dweight;
  c.image_dweight = k->image_dweight;
  c.edge_dweight = k->edge_dweight;
  c.device = k->device;
  c.bf16 = k->compute_dtype == 1;
  c.x3 = k->compute_dtype == 2;
  EG_CHECK((unsigned)k->e_norm <= 2u && (unsigned)k->g_norm <= 2u && (unsigned)k->d_norm <= 2u,
           "e_norm / g_norm / d_norm: 0 = instance, 1 = batch, 2 = None");
  c.e_bn = k->e_norm == 1;
  c.g_bn = k->g_norm == 1;
  c.d_bn = k->d_norm == 1;
  c.e_nonorm = k->e_norm == 2;
  c.g_nonorm = k->g_norm == 2;
  c.d_nonorm = k->d_norm == 2;
  return c;
}

#define EG_API_BEGIN try {
// entry points that enqueue work: always start on lane 0 (the context's own stream; train_step ends there too, and
// an error in the middle of a concurrent iteration may have left another lane active)
#define EG_API_BEGIN_CTX(ctxp) try { api_enter((ctxp)->c);
#define EG_API_END(ctxp)                                  \
  }                                                       \
  catch (const eg::Error& e) {                            \
    if (ctxp) (ctxp)->c.last_error = e.what();            \
    else g_create_error = e.what();                       \
    return e.code;                                        \
  }                                                       \
  catch (const std::exception& e) {                       \
    if (ctxp) (ctxp)->c.last_error = e.what();            \
    else g_create_error = e.what();                       \
    return 99;                                            \
  }                                                       \
  return 0;

extern "C" {

int eg_default_config(eg_config* k) {
  if (!k) return 1;
  *k = eg_config{(unsigned)sizeof(eg_config), 64, 64, 128, 14, 100, 64, 64, 3, 1, 128, 1, 128, 10.f, 10.f, 2e-4f, 1.f, 1.f, 1.f, 0, 0, /* e_norm, g_norm, d_norm = instance */ 0, 0, 0};
  return 0;
}

int eg_variable_count(const eg_config* k) {
  try {
    return (int)enumerate_variables(to_config(k)).size();
  } catch (...) {
    return -1;
  }
}

int eg_variable_info(const eg_config* k, int index, char* name, int name_cap, int* shape, int* ndim, int* trainable,
                     int* init_kind, float* init_param) {
  eg_ctx* none = nullptr;
  EG_API_BEGIN
  auto v = enumerate_variables(to_config(k));
  EG_CHECK(index >= 0 && index < (int)v.size(), "variable index out of range");
  const VarInfo& vi = v[index];
  EG_CHECK((int)vi.name.size() + 1 <= name_cap, "name buffer too small");
  strcpy(name, vi.name.c_str());
  *ndim = (int)vi.shape.size();
  for (size_t i = 0; i < vi.shape.size() && i < 4; ++i) shape[i] = vi.shape[i];
  *trainable = vi.trainable ? 1 : 0;
  *init_kind = vi.init_kind;
  *init_param = vi.init_param;
  EG_API_END(none)
}

int eg_create(const eg_config* k, eg_ctx** out) {
  eg_ctx* none = nullptr;
  eg_ctx* ctx = nullptr;
  try {
    EG_CHECK(k && out, "null argument");
    const Config cfg = to_config(k);   // (first: a layout mismatch is reported as such, with or without a device)
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
      fail(2, "no HIP device available (%s); the engine has no CPU fallback", hipGetErrorString(e));
    ctx = new eg_ctx();
    ctx->c.cfg = cfg;
    build_context(ctx->c);
    *out = ctx;
    return 0;
  } catch (const eg::Error& e) {
    g_create_error = e.what();
    delete ctx;
    return e.code;
  } catch (const std::exception& e) {
    g_create_error = e.what();
    delete ctx;
    return 99;
  }
  (void)none;
}

int eg_destroy(eg_ctx* ctx) {
  if (!ctx) return 0;
  try {
    (void)hipStreamSynchronize(ctx->c.stream);
    comm_destroy(ctx->c);
  } catch (...) {
  }
  delete ctx;
  return 0;
}

const char* eg_last_error(eg_ctx* ctx) { return ctx ? ctx->c.last_error.c_str() : g_create_error.c_str(); }

static bool ends_with(const std::string& s, const std::string& suf) {
  return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}

static float* resolve_var(Ctx& c, const std::string& name, size_t* numel, bool for_set) {
  std::string base = name;
  int which = 0;  // 0 w, 1 ms, 2 mom, 3 grad
  if (ends_with(name, "/RMSProp_1")) {
    base = name.substr(0, name.size() - 10);
    which = 2;
  } else if (ends_with(name, "/RMSProp")) {
    base = name.substr(0, name.size() - 8);
    which = 1;
  } else if (ends_with(name, "/grad")) {
    base = name.substr(0, name.size() - 5);
    which = 3;
  }
  Param* p = P(c, base);
  *numel = p->numel;
  if (which == 0) return p->w;
  EG_CHECK(p->trainable, "%s has no optimizer slot / gradient", base.c_str());
  if (which == 3) {
    EG_CHECK(!for_set, "gradients are read-only");
    EG_CHECK(p->g_set, "no gradient recorded for %s", base.c_str());
    return p->g;
  }
  return which == 1 ? p->ms : p->mom;
}

int eg_set_variable(eg_ctx* ctx, const char* name, const float* host, long long numel) {
  EG_API_BEGIN_CTX(ctx)
  ctx->c.inf_warm = false;
  comm_wait_all(ctx->c);
  size_t n;
  float* d = resolve_var(ctx->c, name, &n, true);
  EG_CHECK((size_t)numel == n, "size mismatch for %s: got %lld, want %zu", name, numel, n);
  EG_HIP(hipMemcpyAsync(d, host, n * 4, hipMemcpyHostToDevice, ctx->c.stream));
  EG_HIP(hipStreamSynchronize(ctx->c.stream));
  ctx->c.sn_valid = false;
  ctx->c.gen_cache_valid[0] = ctx->c.gen_cache_valid[1] = false;
  for (auto& p : ctx->c.params) p->invalidate_derived();
  EG_API_END(ctx)
}

int eg_get_variable(eg_ctx* ctx, const char* name, float* host, long long numel) {
  EG_API_BEGIN_CTX(ctx)
  comm_wait_all(ctx->c);
  size_t n;
  float* d = resolve_var(ctx->c, name, &n, false);
  EG_CHECK((size_t)numel == n, "size mismatch for %s: got %lld, want %zu", name, numel, n);
  EG_HIP(hipMemcpyAsync(host, d, n * 4, hipMemcpyDeviceToHost, ctx->c.stream));
  EG_HIP(hipStreamSynchronize(ctx->c.stream));
  EG_API_END(ctx)
}

int eg_upload_batch(eg_ctx* ctx, const float* images, const float* z, const float* alpha, const float* eps) {
  EG_API_BEGIN_CTX(ctx)
  comm_wait_all(ctx->c);
  Ctx& c = ctx->c;
  const Config& cfg = c.cfg;
  EG_CHECK(images && z && alpha && eps, "null batch pointer");
  const size_t zc = cfg.multiclasses() ? cfg.z_dim + 1 : cfg.z_dim;
  EG_HIP(hipMemcpyAsync(c.d_images, images, (size_t)cfg.B * cfg.H * cfg.W * cfg.c_dim * 4, hipMemcpyHostToDevice,
                        c.stream));
  EG_HIP(hipMemcpyAsync(c.d_z, z, (size_t)cfg.B * zc * 4, hipMemcpyHostToDevice, c.stream));
  EG_HIP(hipMemcpyAsync(c.d_alpha, alpha, (size_t)3 * cfg.B * 4, hipMemcpyHostToDevice, c.stream));
  EG_HIP(hipMemcpyAsync(c.d_eps, eps, 4, hipMemcpyHostToDevice, c.stream));
  c.gen_cache_valid[0] = c.gen_cache_valid[1] = false;
  EG_HIP(hipStreamSynchronize(c.stream));
  EG_API_END(ctx)
}

int eg_substep(eg_ctx* ctx, int k, int apply) {
  EG_API_BEGIN_CTX(ctx)
  ctx->c.inf_warm = false;
  run_substep(ctx->c, k, apply != 0);
  EG_API_END(ctx)
}

int eg_train_step(eg_ctx* ctx) {
  EG_API_BEGIN_CTX(ctx)
  ctx->c.inf_warm = false;
  train_step(ctx->c);
  EG_API_END(ctx)
}

// Pipelined form of upload + train_step + read_losses for the training loop: nothing in it waits for the GPU.  The batch
// is copied into one of two pinned host slots, its host->device copies and the iteration are enqueued behind the
// previous iteration on the engine's stream (6.3 MB at 64x128, batch 64: ~0.15 ms of a ~40 ms iteration), and the
// iteration's losses are copied to a pinned slot behind it.  eg_losses_wait(back) blocks only until THAT iteration has
// finished, so a loop that reads iteration k-1's losses after enqueueing iteration k never lets the GPU run dry.
int eg_train_batch(eg_ctx* ctx, const float* images, const float* z, const float* alpha, const float* eps) {
  EG_API_BEGIN_CTX(ctx)
  Ctx& c = ctx->c;
  c.inf_warm = false;
  const Config& cfg = c.cfg;
  EG_CHECK(images && z && alpha && eps, "null batch pointer");
  const size_t zc = cfg.multiclasses() ? cfg.z_dim + 1 : cfg.z_dim;
  const size_t n_img = (size_t)cfg.B * cfg.H * cfg.W * cfg.c_dim, n_z = (size_t)cfg.B * zc, n_a = (size_t)3 * cfg.B;
  const int slot = (int)(c.feed_count & 1);
  if (!c.feed_host[slot]) {
    // all four per-slot resources or none: a failure half way must not leave a slot that looks initialised
    float *fh = nullptr, *lh = nullptr;
    hipEvent_t e1 = nullptr, e2 = nullptr;
    try {
      EG_HIP(hipHostMalloc((void**)&fh, (n_img + n_z + n_a + 4) * sizeof(float), hipHostMallocDefault));
      EG_HIP(hipHostMalloc((void**)&lh, 64 * sizeof(float), hipHostMallocDefault));
      EG_HIP(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
      EG_HIP(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    } catch (...) {
      if (fh) (void)hipHostFree(fh);
      if (lh) (void)hipHostFree(lh);
      if (e1) (void)hipEventDestroy(e1);
      if (e2) (void)hipEventDestroy(e2);
      throw;
    }
    c.loss_host[slot] = lh;
    c.feed_copied[slot] = e1;
    c.loss_ready[slot] = e2;
    c.feed_host[slot] = fh;
  } else {
    EG_HIP(hipEventSynchronize(c.feed_copied[slot]));   // the copies out of this slot (two feeds ago): long done
  }
  float* h = c.feed_host[slot];
  memcpy(h, images, n_img * 4);
  memcpy(h + n_img, z, n_z * 4);
  memcpy(h + n_img + n_z, alpha, n_a * 4);
  memcpy(h + n_img + n_z + n_a, eps, 4);
  EG_HIP(hipMemcpyAsync(c.d_images, h, n_img * 4, hipMemcpyHostToDevice, c.stream));
  EG_HIP(hipMemcpyAsync(c.d_z, h + n_img, n_z * 4, hipMemcpyHostToDevice, c.stream));
  EG_HIP(hipMemcpyAsync(c.d_alpha, h + n_img + n_z, n_a * 4, hipMemcpyHostToDevice, c.stream));
  EG_HIP(hipMemcpyAsync(c.d_eps, h + n_img + n_z + n_a, 4, hipMemcpyHostToDevice, c.stream));
  EG_HIP(hipEventRecord(c.feed_copied[slot], c.stream));
  c.gen_cache_valid[0] = c.gen_cache_valid[1] = false;
  train_step(c);
  use_lane(c, 0);
  EG_HIP(hipMemcpyAsync(c.loss_host[slot], c.d_losses, L_COUNT * 4, hipMemcpyDeviceToHost, c.stream));
  EG_HIP(hipEventRecord(c.loss_ready[slot], c.stream));
  ++c.feed_count;
  EG_API_END(ctx)
}

int eg_losses_wait(eg_ctx* ctx, int back, float* out12) {
  EG_API_BEGIN
  Ctx& c = ctx->c;
  EG_CHECK(out12 && (back == 0 || back == 1), "eg_losses_wait: out12 != NULL, back in {0, 1}");
  EG_CHECK(c.feed_count > back, "eg_losses_wait: no such iteration (eg_train_batch calls so far: %lld)", c.feed_count);
  const int slot = (int)((c.feed_count - 1 - back) & 1);
  EG_HIP(hipEventSynchronize(c.loss_ready[slot]));
  memcpy(out12, c.loss_host[slot], L_COUNT * 4);
  EG_API_END(ctx)
}

int eg_read_losses(eg_ctx* ctx, float* out12) {
  EG_API_BEGIN_CTX(ctx)
  comm_wait_all(ctx->c);
  EG_HIP(hipMemcpyAsync(out12, ctx->c.d_losses, L_COUNT * 4, hipMemcpyDeviceToHost, ctx->c.stream));
  EG_HIP(hipStreamSynchronize(ctx->c.stream));
  EG_API_END(ctx)
}

int eg_eval_losses(eg_ctx* ctx, const float* alpha, float eps, int bn_updates, const eg_eval_outputs* out) {
  EG_API_BEGIN_CTX(ctx)
  Ctx& c = ctx->c;
  const Config& cfg = c.cfg;
  const int B = cfg.B;
  EG_CHECK(alpha && out && out->losses12, "null argument");
  EG_CHECK(bn_updates >= 0 && bn_updates <= 64, "bn_updates out of range");
  float* d_alpha = c.d_eval + 16 + (size_t)6 * B;
  float* d_eps = d_alpha + (size_t)3 * B;
  EG_HIP(hipMemcpyAsync(d_alpha, alpha, (size_t)3 * B * 4, hipMemcpyHostToDevice, c.stream));
  EG_HIP(hipMemcpyAsync(d_eps, &eps, 4, hipMemcpyHostToDevice, c.stream));
  EG_HIP(hipStreamSynchronize(c.stream));   // `eps` lives on this frame
  EvalResult r;
  eval_losses(c, d_alpha, d_eps, bn_updates, &r);
  auto down = [&](float* host, const float* dev, size_t n) {
    if (host) EG_HIP(hipMemcpyAsync(host, dev, n * 4, hipMemcpyDeviceToHost, c.stream));
  };
  down(out->losses12, c.d_eval, L_COUNT);
  down(out->d_logits, c.d_eval + 16, (size_t)6 * B);
  const size_t half = (size_t)B * cfg.H * (cfg.W / 2) * cfg.c_dim;
  down(out->edge_output, r.edge, half);
  down(out->image_output, r.image, half);
  // a resize to the tensor's own size is the identity (the half image itself); a disabled patch discriminator has no resized
  // tensors (models/edgegan.py:253-274): those outputs are left untouched
  auto rs = [&](int size) { return (cfg.H == size && cfg.W / 2 == size) ? half : (size_t)B * size * size * cfg.c_dim; };
  if (r.resized[0]) down(out->resized_inputs, r.resized[0], rs(cfg.image_dis));
  if (r.resized[1]) down(out->resized_image_output, r.resized[1], rs(cfg.image_dis));
  if (r.resized[2]) down(out->resized_edges, r.resized[2], rs(cfg.edge_dis));
  if (r.resized[3]) down(out->resized_edge_output, r.resized[3], rs(cfg.edge_dis));
  EG_HIP(hipStreamSynchronize(c.stream));
  EG_API_END(ctx)
}

// staging buffers of the inference entry points: grown on demand and kept (no allocation on the steady-state path)
static void infer_reserve(Ctx& c, int nb) {
  const Config& cfg = c.cfg;
  EG_CHECK(nb >= 1 && nb <= 4096, "nb out of range");
  const size_t in_n = (size_t)nb * cfg.H * cfg.W * cfg.c_dim, out_n = in_n / 2;
  if ((size_t)nb <= c.inf_cap) return;
  EG_HIP(hipDeviceSynchronize());   // (runs of the previous, smaller batch may still be in flight on the other slots' lanes)
  (void)hipFree(c.inf_in);
  (void)hipFree(c.inf_cls);
  c.inf_in = nullptr;
  c.inf_cls = nullptr;
  for (int i = 0; i < Ctx::kInferSlots; ++i) {
    (void)hipFree(c.inf_edge[i]);
    (void)hipFree(c.inf_image[i]);
    c.inf_edge[i] = c.inf_image[i] = nullptr;
  }
  c.inf_cap = 0;
  c.inf_nb = 0;   // nothing is resident any more: a failed allocation below must not leave eg_infer_run runnable
  EG_HIP(hipMalloc(&c.inf_in, in_n * 4));
  for (int i = 0; i < Ctx::kInferSlots; ++i) {
    EG_HIP(hipMalloc(&c.inf_edge[i], out_n * 4));
    EG_HIP(hipMalloc(&c.inf_image[i], out_n * 4));
  }
  EG_HIP(hipMalloc(&c.inf_cls, (size_t)nb * 4));
  if (!c.inf_eps) EG_HIP(hipMalloc(&c.inf_eps, 8));
  c.inf_cap = nb;
}

int eg_infer_upload(eg_ctx* ctx, const float* images, const int* classes, int nb, const float* eps2) {
  EG_API_BEGIN_CTX(ctx)
  Ctx& c = ctx->c;
  const Config& cfg = c.cfg;
  EG_CHECK(images && eps2, "null argument");
  EG_CHECK(!cfg.multiclasses() || classes, "classes required");
  infer_reserve(c, nb);
  const size_t in_n = (size_t)nb * cfg.H * cfg.W * cfg.c_dim;
  EG_HIP(hipDeviceSynchronize());   // runs still in flight on any slot read the resident batch this call replaces
  EG_HIP(hipMemcpyAsync(c.inf_in, images, in_n * 4, hipMemcpyHostToDevice, c.stream));
  EG_HIP(hipMemcpyAsync(c.inf_eps, eps2, 8, hipMemcpyHostToDevice, c.stream));
  if (classes) EG_HIP(hipMemcpyAsync(c.inf_cls, classes, (size_t)nb * 4, hipMemcpyHostToDevice, c.stream));
  c.inf_nb = nb;
  EG_HIP(hipStreamSynchronize(c.stream));
  EG_API_END(ctx)
}

int eg_infer_run(eg_ctx* ctx) {
  if (ctx) ctx->c.inf_entering = true;   // (api_enter: this entry point does not drain the runs already in flight)
  EG_API_BEGIN_CTX(ctx)
  Ctx& c = ctx->c;
  EG_CHECK(c.inf_nb > 0, "eg_infer_run without eg_infer_upload");
  // slot of this call: its lanes and its output buffers.  The slots' calls share nothing that is written (engine.h), so up to
  // inf_slots consecutive calls are in flight at once; a slot's next call follows its previous one in stream order.
  const int nslots = (c.lanes_on && !c.prof_on) ? c.inf_slots : 1;
  const int slot = (int)(c.inf_runs++ % nslots);
  const int main_lane = nslots >= 4 ? slot : 2 * slot, side_lane = nslots >= 4 ? -1 : main_lane + 1;
  c.inf_last_slot = slot;
  infer(c, c.inf_in, c.inf_cls, c.inf_nb, c.inf_eps, c.inf_edge[slot], c.inf_image[slot], main_lane, side_lane);
  c.inf_inflight = true;
  if (!c.inf_warm) {   // the first run after the weights changed builds their derived copies on ITS lanes: finish it before another slot reads them
    EG_HIP(hipDeviceSynchronize());
    c.inf_warm = true;
  }
  EG_API_END(ctx)
}

int eg_infer_download(eg_ctx* ctx, float* edge_out, float* image_out) {
  EG_API_BEGIN_CTX(ctx)
  Ctx& c = ctx->c;
  const Config& cfg = c.cfg;
  EG_CHECK(c.inf_nb > 0 && edge_out && image_out, "eg_infer_download: nothing resident / null argument");
  const size_t out_n = (size_t)c.inf_nb * cfg.H * (cfg.W / 2) * cfg.c_dim;
  EG_HIP(hipDeviceSynchronize());   // every slot's lanes: the latest run (and all before it) have finished
  EG_HIP(hipMemcpyAsync(edge_out, c.inf_edge[c.inf_last_slot], out_n * 4, hipMemcpyDeviceToHost, c.stream));
  EG_HIP(hipMemcpyAsync(image_out, c.inf_image[c.inf_last_slot], out_n * 4, hipMemcpyDeviceToHost, c.stream));
  EG_HIP(hipStreamSynchronize(c.stream));
  EG_API_END(ctx)
}

int eg_infer(eg_ctx* ctx, const float* images, const int* classes, int nb, const float* eps2, float* edge_out,
             float* image_out) {
  int rc = eg_infer_upload(ctx, images, classes, nb, eps2);
  if (rc == 0) rc = eg_infer_run(ctx);
  if (rc == 0) rc = eg_infer_download(ctx, edge_out, image_out);
  return rc;
}

int eg_sync(eg_ctx* ctx) {
  EG_API_BEGIN_CTX(ctx)
  comm_wait_all(ctx->c);
  EG_HIP(hipStreamSynchronize(ctx->c.stream));
  EG_HIP(hipDeviceSynchronize());   // every lane, leaf and exchange stream of this device: the host-side "everything is done"
  EG_API_END(ctx)
}

int eg_comm_unique_id(eg_ctx* ctx, char* out128) {
  EG_API_BEGIN
  comm_unique_id(ctx->c, out128);
  EG_API_END(ctx)
}
int eg_comm_init(eg_ctx* ctx, const char* id128, int rank, int nranks) {
  EG_API_BEGIN
  comm_init(ctx->c, id128, rank, nranks);
  EG_API_END(ctx)
}

int eg_comm_info(eg_ctx* ctx, int* nranks, int* rccl_version, char* libpath, int libpath_cap, long long* allreduce_calls,
                 long long* allreduce_bytes) {
  EG_API_BEGIN
  EG_CHECK(nranks && rccl_version && allreduce_calls && allreduce_bytes, "null argument");
  *nranks = ctx->c.nranks;
  comm_info(ctx->c, rccl_version, libpath, libpath_cap, allreduce_calls, allreduce_bytes);
  EG_API_END(ctx)
}

int eg_comm_reduce_scalars(eg_ctx* ctx, float* inout, int n, int op) {
  EG_API_BEGIN
  EG_CHECK(inout, "null argument");
  comm_reduce_scalars(ctx->c, inout, n, op);
  EG_API_END(ctx)
}

int eg_comm_timing(eg_ctx* ctx, int enable) {
  EG_API_BEGIN
  EG_CHECK(ctx, "null context");
  ctx->c.comm_timing = enable != 0;
  EG_API_END(ctx)
}
int eg_comm_timing_read(eg_ctx* ctx, float* ms, int* group, long long* bytes, int cap, int* n) {
  EG_API_BEGIN
  EG_CHECK(ctx && ms && group && bytes && n && cap >= 0, "null argument");
  *n = comm_timing_read(ctx->c, ms, group, bytes, cap);
  EG_API_END(ctx)
}

int eg_profile_begin(eg_ctx* ctx) {
  EG_API_BEGIN
  Ctx& c = ctx->c;
  EG_HIP(hipStreamSynchronize(c.stream));
  for (auto& r : c.prof) {
    c.ev_pool.push_back(r.a);
    c.ev_pool.push_back(r.b);
  }
  c.prof.clear();
  c.prof_on = true;
  EG_API_END(ctx)
}

int eg_profile_end(eg_ctx* ctx, char* buf, int cap) {
  EG_API_BEGIN
  Ctx& c = ctx->c;
  EG_HIP(hipStreamSynchronize(c.stream));
  c.prof_on = false;
  struct Agg {
    long long n = 0;
    double ms = 0, flops = 0;
  };
  std::map<std::string, Agg> agg;
  for (auto& r : c.prof) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
    Agg& a = agg[r.name];
    a.n += 1;
    a.ms += ms;
    a.flops += r.flops;
  }
  std::ostringstream os;
  os << "{\"kernels\":[";
  bool first = true;
  for (auto& kv : agg) {
    if (!first) os << ",";
    first = false;
    os << "{\"name\":\"" << kv.first << "\",\"launches\":" << kv.second.n << ",\"ms\":" << kv.second.ms
       << ",\"flops\":" << kv.second.flops << "}";
  }
  os << "]}";
  const std::string s = os.str();
  EG_CHECK((int)s.size() + 1 <= cap, "profile buffer too small");
  strcpy(buf, s.c_str());
  EG_API_END(ctx)
}

// one CSV line per GEMM-shaped launch recorded since eg_profile_begin (call BEFORE eg_profile_end, which also works
// afterwards: the records are kept until the next eg_profile_begin)
int eg_profile_layers(eg_ctx* ctx, char* buf, long long cap) {
  EG_API_BEGIN
  Ctx& c = ctx->c;
  EG_HIP(hipStreamSynchronize(c.stream));
  std::ostringstream os;
  os << "label,kernel,M,N,C,taps,nphase,split,us,gflop\n";
  for (auto& r : c.prof) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
    os << r.label << "," << '"' << r.name << '"' << "," << (r.shape.empty() ? ",,,,," : r.shape) << "," << ms * 1e3 << ","
       << r.flops * 1e-9 << "\n";
  }
  const std::string s = os.str();
  EG_CHECK((long long)s.size() + 1 <= cap, "profile buffer too small: need %zu bytes", s.size() + 1);
  memcpy(buf, s.c_str(), s.size() + 1);
  EG_API_END(ctx)
}

int eg_mfma_probe(eg_ctx* ctx, int iters, int random_operands, double* tflops) {
  EG_API_BEGIN
  EG_CHECK(tflops && iters > 0, "bad argument");
  int dev = 0, cus = 0;
  EG_HIP(hipGetDevice(&dev));
  EG_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  EG_HIP(hipStreamSynchronize(ctx->c.stream));
  float* scratch = nullptr;
  EG_HIP(hipMalloc(&scratch, (size_t)cus * 512 * sizeof(float)));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  float ms = 0.f;
  try {
    EG_HIP(hipEventCreate(&e0));
    EG_HIP(hipEventCreate(&e1));
    launch_mfma_probe(scratch, cus, 64, random_operands, ctx->c.stream);   // warm-up
    EG_HIP(hipEventRecord(e0, ctx->c.stream));
    launch_mfma_probe(scratch, cus, iters, random_operands, ctx->c.stream);
    EG_HIP(hipEventRecord(e1, ctx->c.stream));
    EG_HIP(hipEventSynchronize(e1));
    EG_HIP(hipEventElapsedTime(&ms, e0, e1));
  } catch (...) {
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(scratch);
    throw;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  EG_HIP(hipFree(scratch));
  *tflops = 2.0 * 32 * 32 * 16 * 16.0 * iters * 8 * cus / (ms * 1e9);
  EG_API_END(ctx)
}

int eg_memory_stats(eg_ctx* ctx, long long* peak, long long* reserved) {
  EG_API_BEGIN
  size_t pk = ctx->c.arena.peak(), rs = ctx->c.arena.reserved();   // active lane + the parked ones
  for (int i = 0; i < Ctx::kLanes; ++i) {
    pk += ctx->c.lanes[i].arena.peak();
    rs += ctx->c.lanes[i].arena.reserved();
  }
  *peak = (long long)pk;
  *reserved = (long long)rs;
  EG_API_END(ctx)
}

// -------------------------------------------------------------------------------------------------
// per-op test scaffolding
// -------------------------------------------------------------------------------------------------
namespace {
struct OpIO {
  Ctx& c;
  eg_ctx* ctx;
  int n_in;
  const float* const* in;
  const long long* in_n;
  int n_out;
  float* const* out;
  const long long* out_n;
  bool has(int i) const { return i < n_in && in_n[i] > 0 && in[i]; }
  float* up(int i) {
    EG_CHECK(has(i), "test op input %d missing", i);
    float* d = c.arena.allocf((size_t)in_n[i]);
    EG_HIP(hipMemcpyAsync(d, in[i], (size_t)in_n[i] * 4, hipMemcpyHostToDevice, c.stream));
    return d;
  }
  T tensor(int i, int n, int h, int w, int ch, bool grad) {
    EG_CHECK((long long)n * h * w * ch == in_n[i], "test op input %d has %lld elements, expected %lld", i, in_n[i],
             (long long)n * h * w * ch);
    T t = wrap_tensor(c, up(i), n, h, w, ch);
    t->needs_grad = grad;
    return t;
  }
  Param* param(int i, bool want) {
    ctx->tmp_params.emplace_back();
    Param* p = &ctx->tmp_params.back();
    p->name = "tmp" + std::to_string(i);
    p->numel = (size_t)in_n[i];
    p->w = up(i);
    p->g = c.arena.allocf(p->numel);
    p->want_grad = want;
    return p;
  }
  void seed(T y, int i) {
    EG_CHECK((long long)y->numel() == in_n[i], "cotangent %d size %lld vs output %zu", i, in_n[i], y->numel());
    seed_grad(c, y);
    EG_HIP(hipMemcpyAsync(y->g, in[i], y->numel() * 4, hipMemcpyHostToDevice, c.stream));
  }
  void down(int o, const float* d, size_t n) {
    if (o >= n_out || !out[o] || out_n[o] == 0) return;
    EG_CHECK((size_t)out_n[o] == n, "test op output %d has %lld elements, expected %zu", o, out_n[o], n);
    EG_HIP(hipMemcpyAsync(out[o], d, n * 4, hipMemcpyDeviceToHost, c.stream));
  }
  void down_grad(int o, T t) {
    if (o >= n_out || !out[o] || out_n[o] == 0) return;
    EG_CHECK(t->g_set, "no gradient reached test op input for output %d", o);
    down(o, t->g, t->numel());
  }
  void down_pgrad(int o, Param* p) {
    if (o >= n_out || !out[o] || out_n[o] == 0) return;
    EG_CHECK(p->g_set, "no gradient reached test op parameter for output %d", o);
    down(o, p->g, p->numel);
  }
};
}  // namespace

int eg_test_op(eg_ctx* ctx, const char* op_, int n_in, const float* const* inputs, const long long* in_numel,
               int n_ip, const int* ip, int n_out, float* const* outputs, const long long* out_numel) {
  EG_API_BEGIN_CTX(ctx)
  comm_wait_all(ctx->c);
  Ctx& c = ctx->c;
  reset_step(c);
  ctx->tmp_params.clear();
  OpIO io{c, ctx, n_in, inputs, in_numel, n_out, outputs, out_numel};
  const std::string op = op_;
  auto need_ip = [&](int n) { EG_CHECK(n_ip >= n, "op %s needs %d int params", op_, n); };
  if (op == "conv2d") {  // ip: NB,H,W,C,N,k,stride,pad ; in: x,w,b?,dy? ; out: y,dx,dw,db
    need_ip(8);
    T x = io.tensor(0, ip[0], ip[1], ip[2], ip[3], io.has(3));
    Param* w = io.param(1, io.has(3));
    Param* b = io.has(2) ? io.param(2, io.has(3)) : nullptr;
    T y = conv2d(c, x, wref_param(c, w, ip[5] * ip[5], ip[3], ip[4]), b, ip[5], ip[6], ip[7]);
    io.down(0, y->d, y->numel());
    if (io.has(3)) {
      io.seed(y, 3);
      c.tape.backward();
      io.down_grad(1, x);
      io.down_pgrad(2, w);
      if (b) io.down_pgrad(3, b);
    }
  } else if (op == "deconv2d") {  // ip: NB,H,W,Cin,Cout ; in: x,w,b,dy? ; out: y,dx,dw,db
    need_ip(5);
    T x = io.tensor(0, ip[0], ip[1], ip[2], ip[3], io.has(3));
    Param* w = io.param(1, io.has(3));
    Param* b = io.param(2, io.has(3));
    T y = deconv2d(c, x, wref_param(c, w, 25, ip[4], ip[3]), b, ip[4]);
    io.down(0, y->d, y->numel());
    if (io.has(3)) {
      io.seed(y, 3);
      c.tape.backward();
      io.down_grad(1, x);
      io.down_pgrad(2, w);
      io.down_pgrad(3, b);
    }
  } else if (op == "in_act" || op == "act") {  // ip: NB,H,W,C,act ; in: x,dy? ; out: y,dx
    need_ip(5);
    T x = io.tensor(0, ip[0], ip[1], ip[2], ip[3], io.has(1));
    T y = op == "act" ? act(c, x, ip[4]) : in_act(c, x, ip[4]);
    io.down(0, y->d, y->numel());
    if (io.has(1)) {
      io.seed(y, 1);
      c.tape.backward();
      io.down_grad(1, x);
    }
  } else if (op == "in_tangent") {  // ip: NB,H,W,C,act ; in: x,tx,dty ; out: y,ty,dtx,dx
    need_ip(5);
    T x = io.tensor(0, ip[0], ip[1], ip[2], ip[3], true);
    T tx = io.tensor(1, ip[0], ip[1], ip[2], ip[3], true);
    T st;
    T y = in_act(c, x, ip[4], &st);
    T ty = in_act_tangent(c, tx, x, y, st, ip[4]);
    io.down(0, y->d, y->numel());
    io.down(1, ty->d, ty->numel());
    if (io.has(2)) {
      io.seed(ty, 2);
      c.tape.backward();
      io.down_grad(2, tx);
      io.down_grad(3, x);
    }
  } else if (op == "bn_relu") {  // ip: NB,H,W,C ; in: x,gamma,beta,dy? ; out: y,dx,dgamma,dbeta
    need_ip(4);
    T x = io.tensor(0, ip[0], ip[1], ip[2], ip[3], io.has(3));
    Param* g = io.param(1, io.has(3));
    Param* b = io.param(2, io.has(3));
    T y = bn_relu(c, x, g, b, nullptr, nullptr, 0);
    io.down(0, y->d, y->numel());
    if (io.has(3)) {
      io.seed(y, 3);
      c.tape.backward();
      io.down_grad(1, x);
      io.down_pgrad(2, g);
      io.down_pgrad(3, b);
    }
  } else if (op == "bn_act") {  // ip: NB,H,W,C,act ; in: x,gamma,beta,dy? ; out: y,dx,dgamma,dbeta
    need_ip(5);
    T x = io.tensor(0, ip[0], ip[1], ip[2], ip[3], io.has(3));
    Param* g = io.param(1, io.has(3));
    Param* b = io.param(2, io.has(3));
    T y = bn_act(c, x, g, b, nullptr, nullptr, ip[4], 0);
    io.down(0, y->d, y->numel());
    if (io.has(3)) {
      io.seed(y, 3);
      c.tape.backward();
      io.down_grad(1, x);
      io.down_pgrad(2, g);
      io.down_pgrad(3, b);
    }
  } else if (op == "bn_tangent") {  // ip: NB,H,W,C,act ; in: x,gamma,beta,tx,dty ; out: y,ty,dtx,dx,dgamma
    need_ip(5);
    T x = io.tensor(0, ip[0], ip[1], ip[2], ip[3], true);
    Param* g = io.param(1, true);
    Param* b = io.param(2, true);
    T tx = io.tensor(3, ip[0], ip[1], ip[2], ip[3], true);
    T st;
    T y = bn_act(c, x, g, b, nullptr, nullptr, ip[4], 0, &st);
    T ty = bn_act_tangent(c, tx, x, y, st, g, ip[4]);
    io.down(0, y->d, y->numel());
    io.down(1, ty->d, ty->numel());
    if (io.has(4)) {
      io.seed(ty, 4);
      c.tape.backward();
      io.down_grad(2, tx);
      io.down_grad(3, x);
      io.down_pgrad(4, g);
    }
  } else if (op == "resize" || op == "pool2" || op == "gmean" || op == "minmax") {
    need_ip(5);  // ip: NB,H,W,C,(size | pre_relu | unused) ; in: x,dy? ; out: y,dx
    T x = io.tensor(0, ip[0], ip[1], ip[2], ip[3], io.has(1));
    T y = op == "resize" ? resize(c, x, ip[4])
          : op == "pool2" ? pool2(c, x)
          : op == "gmean" ? gmean(c, x, ip[4] != 0)
                          : minmax(c, x);
    io.down(0, y->d, y->numel());
    if (io.has(1)) {
      io.seed(y, 1);
      c.tape.backward();
      io.down_grad(1, x);
    }
  } else if (op == "prelu") {  // ip: n ; in: x,leak,dy? ; out: y,dx,dleak
    need_ip(1);
    T x = io.tensor(0, 1, 1, 1, ip[0], io.has(2));
    Param* lk = io.param(1, io.has(2));
    T y = prelu(c, x, lk, 0.f);
    io.down(0, y->d, y->numel());
    if (io.has(2)) {
      io.seed(y, 2);
      c.tape.backward();
      io.down_grad(1, x);
      io.down_pgrad(2, lk);
    }
  } else if (op == "gate") {  // ip: n ; in: h,rg,img,dy ; out: y,dh,drg,dimg
    need_ip(1);
    T h = io.tensor(0, 1, 1, 1, ip[0], true);
    T rg = io.tensor(1, 1, 1, 1, ip[0], true);
    T im = io.tensor(2, 1, 1, 1, ip[0], true);
    T y = gate(c, h, rg, im);
    io.down(0, y->d, y->numel());
    if (io.has(3)) {
      io.seed(y, 3);
      c.tape.backward();
      io.down_grad(1, h);
      io.down_grad(2, rg);
      io.down_grad(3, im);
    }
  } else if (op == "linear1") {  // ip: NB,K ; in: x,w,b,dy? ; out: y,dx,dw,db
    need_ip(2);
    T x = io.tensor(0, ip[0], 1, 1, ip[1], io.has(3));
    Param* w = io.param(1, io.has(3));
    Param* b = io.param(2, io.has(3));
    T y = linear1(c, x, w, b);
    io.down(0, y->d, y->numel());
    if (io.has(3)) {
      io.seed(y, 3);
      c.tape.backward();
      io.down_grad(1, x);
      io.down_pgrad(2, w);
      io.down_pgrad(3, b);
    }
  } else if (op == "sn") {  // ip: K,Co ; in: W,u,dWbar? ; out: Wbar,dW,sigma
    need_ip(2);
    const int K = ip[0], Co = ip[1];
    SNDesc d{};
    float* W = io.up(0);
    d.W = W;
    d.u = io.up(1);
    d.Wbar = c.arena.allocf((size_t)K * Co);
    d.t = c.arena.allocf(K);
    d.dv = c.arena.allocf(K);
    d.sv = c.arena.allocf(sn_sv_floats(Co));
    d.scal = c.arena.allocf(SN_SCAL_FLOATS);
    d.dW = c.arena.allocf((size_t)K * Co);
    d.dWbar = io.has(2) ? io.up(2) : nullptr;
    d.K = K;
    d.Co = Co;
    d.acc_dW = 0;
    SNDesc* tab = (SNDesc*)c.arena.alloc(sizeof(SNDesc));
    EG_HIP(hipMemcpyAsync(tab, &d, sizeof(SNDesc), hipMemcpyHostToDevice, c.stream));
    EG_HIP(hipStreamSynchronize(c.stream));
    launch_sn_fwd(tab, 1, K, Co, c.stream);
    io.down(0, d.Wbar, (size_t)K * Co);
    io.down(2, d.scal + 2, 1);
    if (io.has(2)) {
      launch_sn_bwd(tab, 1, K, Co, c.stream);
      io.down(1, d.dW, (size_t)K * Co);
    }
  } else if (op == "acgan") {  // ip: B,NC,mode ; in: logits,labels ; out: loss,dlogits
    need_ip(3);
    float* lg = io.up(0);
    float* lab = io.up(1);
    float* loss = c.arena.allocf(1);
    float* dl = c.arena.allocf((size_t)ip[0] * ip[1]);
    launch_acgan_loss(lg, lab, 1, loss, dl, ip[0], ip[1], ip[2], c.scratch, c.stream);
    io.down(0, loss, 1);
    io.down(1, dl, (size_t)ip[0] * ip[1]);
  } else if (op == "gp") {  // ip: B,per ; in: g ; out: v,norms,gp
    need_ip(2);
    float* g = io.up(0);
    float* v = c.arena.allocf((size_t)ip[0] * ip[1]);
    float* nr = c.arena.allocf(ip[0]);
    float* gp = c.arena.allocf(1);
    launch_gp_v(g, v, nr, gp, ip[0], (size_t)ip[1], c.cfg.lambda_gp, c.stream);
    io.down(0, v, (size_t)ip[0] * ip[1]);
    io.down(1, nr, ip[0]);
    io.down(2, gp, 1);
  } else if (op == "rmsprop") {  // ip: n ; in: w,g,ms,mom,lr ; out: w,ms,mom
    need_ip(1);
    float* w = io.up(0);
    float* g = io.up(1);
    float* ms = io.up(2);
    float* mom = io.up(3);
    launch_rmsprop(w, g, ms, mom, (size_t)ip[0], inputs[4][0], 1.f, c.stream);
    io.down(0, w, ip[0]);
    io.down(1, ms, ip[0]);
    io.down(2, mom, ip[0]);
  } else if (op == "mru_tail") {
    // the tail of an MRU unit (conv.py:217-243) through the fused forms of round 5: Conv_1 + PReLU (second store of the
    // convolution's epilogue, or the PReLU pass summing K-split slabs), Conv_2 read by pool2_add as slabs, the next unit's
    // prelu(h) as pool2_add's second store; backward: K-split input-gradient slabs summed by prelu_bwd
    // ip: NB,H,W,C1,C2 ; in: x,w1,b1,leak1,w2,b2,bp,leak2,dh,dhp ; out: h,hp,dx,dw1,db1,dleak1,dw2,db2,dbp,dleak2
    // ip[5] (optional) = 1: the parameters are frozen, as the classifier's are in the generator sub-steps -- only dx and dbp are
    // formed, and Conv_1's PReLU backward may ride in Conv_2's input-gradient epilogue (ConvArgs::epi_act = ACT_PRELU_X)
    need_ip(5);
    const int NB = ip[0], H = ip[1], W = ip[2], C1 = ip[3], C2 = ip[4];
    const bool bw = io.has(8), pg = bw && !(n_ip > 5 && ip[5]);
    T x = io.tensor(0, NB, H, W, C1, bw);
    Param* w1 = io.param(1, pg);
    Param* b1 = io.param(2, pg);
    Param* l1 = io.param(3, pg);
    Param* w2 = io.param(4, pg);
    Param* b2 = io.param(5, pg);
    T bp = io.tensor(6, NB, H / 2, W / 2, C2, bw);
    Param* l2 = io.param(7, pg);
    c.prelu_plane_only = c.cfg.bf16;   // (as classifier_forward does)
    T c1 = conv2d(c, x, wref_param(c, w1, 9, C1, C2), b1, 3, 1, PAD_SAME, nullptr, 0, l1);
    T hp = nullptr;
    T h = conv2d_pool2_add(c, c1, wref_param(c, w2, 9, C2, C2), b2, 3, bp, l2, &hp);   // (pooled in Conv_2's epilogue where its kernel can)
    io.down(0, h->d, h->numel());
    io.down(1, hp->d, hp->numel());
    if (bw) {
      io.seed(h, 8);
      io.seed(hp, 9);
      c.tape.backward();
      io.down_grad(2, x);
      if (pg) {
        io.down_pgrad(3, w1);
        io.down_pgrad(4, b1);
        io.down_pgrad(5, l1);
        io.down_pgrad(6, w2);
        io.down_pgrad(7, b2);
      }
      io.down_grad(8, bp);
      if (pg) io.down_pgrad(9, l2);
    }
  } else if (op == "mru_gate") {
    // the head of an MRU unit (conv.py:185-209): prelu(h), the update gate as image rows + hidden rows (the hidden-state
    // convolution's K-split slabs summed by the min-max pass), gate + PReLU, and a convolution behind it whose input
    // gradient may reach gate_prelu's backward as slabs
    // ip: NB,H,W,Ch,N ; in: h,xin,leak0,wh,wx,bg,img,leak1,w1,dy ; out: y,dh,dxin,dimg,dleak0,dwh,dwx,dbg,dleak1,dw1
    need_ip(5);
    const int NB = ip[0], H = ip[1], W = ip[2], Ch = ip[3], N = ip[4];
    const bool bw = io.has(9);
    T h = io.tensor(0, NB, H, W, Ch, bw);
    T xin = io.tensor(1, NB, H, W, 3, bw);
    Param* l0 = io.param(2, bw);
    Param* wh = io.param(3, bw);
    Param* wx = io.param(4, bw);
    Param* bg = io.param(5, bw);
    T img = io.tensor(6, NB, H, W, Ch, bw);
    Param* l1 = io.param(7, bw);
    Param* w1 = io.param(8, bw);
    T hp = prelu(c, h, l0, 0.f);
    T rg = conv2d(c, xin, wref_param(c, wx, 9, 3, Ch), bg, 3, 1, PAD_SAME);
    c.feeds_in = FEED_MINMAX;
    conv2d(c, hp, wref_param(c, wh, 9, Ch, Ch), nullptr, 3, 1, PAD_SAME, rg);
    rg = minmax(c, rg, 0.2f);
    T hin = gate_prelu(c, h, rg, img, l1, c.cfg.bf16);
    T y = conv2d(c, hin, wref_param(c, w1, 9, Ch, N), nullptr, 3, 1, PAD_SAME);
    io.down(0, y->d, y->numel());
    if (bw) {
      io.seed(y, 9);
      c.tape.backward();
      io.down_grad(1, h);
      io.down_grad(2, xin);
      io.down_grad(3, img);
      io.down_pgrad(4, l0);
      io.down_pgrad(5, wh);
      io.down_pgrad(6, wx);
      io.down_pgrad(7, bg);
      io.down_pgrad(8, l1);
      io.down_pgrad(9, w1);
    }
  } else if (op == "act_conv") {
    // x -> {in_act(LeakyReLU) | pool2} -> convolution: the convolution's input gradient, first writer of the middle tensor's
    // gradient, may leave K-split slabs that in_bwd / pool2_bwd sum while loading
    // ip: NB,H,W,C,N,k,stride,kind(0 in_act, 1 pool2) ; in: x,w,dy ; out: y,dx,dw
    need_ip(8);
    const bool bw = io.has(2);
    T x = io.tensor(0, ip[0], ip[1], ip[2], ip[3], bw);
    Param* w = io.param(1, bw);
    T m = ip[7] == 0 ? in_act(c, x, ACT_LRELU) : pool2(c, x);
    T y = conv2d(c, m, wref_param(c, w, ip[5] * ip[5], ip[3], ip[4]), nullptr, ip[5], ip[6], PAD_SAME);
    io.down(0, y->d, y->numel());
    if (bw) {
      io.seed(y, 2);
      c.tape.backward();
      io.down_grad(1, x);
      io.down_pgrad(2, w);
    }
  } else if (op == "d_head") {
    // d_conv_0 + LeakyReLU (fused into the convolution where its kernel can) followed by d_conv_1 (discriminator.py:61-66): the
    // second convolution's input gradient is the ONLY writer of the activated tensor's gradient, so the activation's backward
    // rides in its epilogue when the kernel that runs reports CONV_CAP_EPI_ACT (un-split halo kernel) and is the in-place
    // pass of conv2d's backward otherwise (K-split / 64x64 kernels) -- both must give the same gradient
    // ip: NB,H,W,N1,N2 ; in: x,w0,w1,dy ; out: y,dx,dw0,dw1
    need_ip(5);
    const bool bw = io.has(3);
    T x = io.tensor(0, ip[0], ip[1], ip[2], 3, bw);
    Param* w0 = io.param(1, bw);
    Param* w1 = io.param(2, bw);
    T a0 = conv2d(c, x, wref_param(c, w0, 16, 3, ip[3]), nullptr, 4, 2, PAD_SAME, nullptr, ACT_LRELU);
    T y = conv2d(c, a0, wref_param(c, w1, 16, ip[3], ip[4]), nullptr, 4, 2, PAD_SAME);
    io.down(0, y->d, y->numel());
    if (bw) {
      io.seed(y, 3);
      c.tape.backward();
      io.down_grad(1, x);
      io.down_pgrad(2, w0);
      io.down_pgrad(3, w1);
    }
  } else if (op == "fault_mid_backward" && getenv("EG_TEST_HOOKS")) {   // (only where the test suite asks for it)
    // test hook (tests/test_step_gpu.py): leave the context the way an exception in the middle of a concurrent
    // iteration's backward sweep does -- another lane active, its stream pointing at the lane's leaf stream with
    // un-joined work, a queued second-stage reduction of the abandoned sweep -- and fail.  The NEXT call has to recover
    // (model.cpp: api_enter).
    use_lane(c, 2);
    if (c.lanes[2].leaf) c.stream = c.lanes[2].leaf;
    c.leaf_dirty = true;
    fail(7, "injected fault: lane %d active, leaf work un-joined", c.cur_lane);
  } else {
    fail(1, "unknown test op '%s'", op_);
  }
  EG_HIP(hipStreamSynchronize(c.stream));
  EG_API_END(ctx)
}

int eg_net_forward(eg_ctx* ctx, const char* net_, const float* input, int nb, int h, int w, int ch, float eps,
                   float* out, long long out_numel) {
  EG_API_BEGIN_CTX(ctx)
  comm_wait_all(ctx->c);
  Ctx& c = ctx->c;
  reset_step(c);
  for (int g = 0; g < G_COUNT; ++g) set_want_grad(c, g, false);
  const std::string net = net_;
  const size_t n = (size_t)nb * h * w * ch;
  float* d = c.arena.allocf(n);
  EG_HIP(hipMemcpyAsync(d, input, n * 4, hipMemcpyHostToDevice, c.stream));
  T x = wrap_tensor(c, d, nb, h, w, ch);
  T y = nullptr;
  if (net == "G1" || net == "G2") {
    y = generator_forward(c, net == "G1" ? G_G1 : G_G2, x, 0);
  } else if (net == "D" || net == "D_patch2" || net == "D_patch3") {
    y = disc_forward(c, net == "D" ? G_D : (net == "D_patch2" ? G_DP2 : G_DP3), x, nullptr, nullptr);
  } else if (net == "E") {
    float* de = c.arena.allocf(1);
    EG_HIP(hipMemcpyAsync(de, &eps, 4, hipMemcpyHostToDevice, c.stream));
    y = encoder_forward(c, x, de);
  } else if (net == "D2") {
    EG_CHECK(c.cfg.multiclasses(), "classifier absent without classes");
    y = classifier_forward(c, x);
  } else {
    fail(1, "unknown net '%s'", net_);
  }
  EG_CHECK((long long)y->numel() == out_numel, "output has %zu elements, caller expects %lld", y->numel(),
           out_numel);
  EG_HIP(hipMemcpyAsync(out, y->d, y->numel() * 4, hipMemcpyDeviceToHost, c.stream));
  EG_HIP(hipStreamSynchronize(c.stream));
  EG_API_END(ctx)
}

}  // extern "C"
