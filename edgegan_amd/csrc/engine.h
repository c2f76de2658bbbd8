// EdgeGAN MI355X engine -- context, tape ops and network/sub-step entry points (host side).
#pragma once
#include <deque>

#include "common.h"
#include "kernels.h"

namespace eg {

struct Config {
  int B = 64, H = 64, W = 128;
  int num_classes = 14;  // 0 => --nomulticlasses
  int z_dim = 100, gf = 64, df = 64, c_dim = 3;
  bool use_image_dis = true, use_edge_dis = true;   // train.py:61-67: a patch discriminator that is off is never built
  int image_dis = 128, edge_dis = 128;
  float lambda_gp = 10.f, zl_weight = 10.f, lr = 2e-4f;
  float joint_dweight = 1.f, image_dweight = 1.f, edge_dweight = 1.f;
  int device = 0;
  bool bf16 = false;  // dense contractions on the bf16 MFMA (operands rounded, fp32 accumulate / storage)
  bool x3 = false;    // fp32 contractions as three-limb bf16 splits on the bf16 MFMA (exact limb products, fp32 accumulate)
  bool e_bn = false, g_bn = false, d_bn = false;   // E_norm / G_norm / D_norm == 'batch' (train.py:54-59; default 'instance')
  bool e_nonorm = false, g_nonorm = false, d_nonorm = false;   // ... == None: the block's norm is the identity (normalization.py:26-27)
  bool multiclasses() const { return num_classes > 0; }
  int g_in() const { return z_dim + (multiclasses() ? num_classes : 0); }
};

enum Feeds { FEED_NONE = 0, FEED_IN = 1, FEED_FLAT = 2, FEED_MINMAX = 3 };   // Ctx::feeds_in
enum GroupId { G_G1 = 0, G_G2, G_D2, G_D, G_DP2, G_DP3, G_E, G_COUNT };

// One network's variables in flat HBM buffers: [live trainables | dead trainables | non-trainables].
// RMSProp and the DP all-reduce run over the live prefix in a single launch / collective.
struct NetGroup {
  std::string prefix;
  std::vector<Param*> params;
  float *w = nullptr, *g = nullptr, *ms = nullptr, *mom = nullptr;
  size_t n_live = 0, n_train = 0, n_total = 0;
  size_t n_train_hint = 0;   // G2 only: its trainable extent, known before G1's gradient buffer (which hosts G2's) is allocated
};

struct SNWeight;

// Effective weights of one conv/deconv/linear for the current sub-step.
struct WRef {
  const float* w = nullptr;  // [T][R][Cc]
  float* g = nullptr;        // gradient target, same layout
  bool* g_set = nullptr;
  bool want_grad = false;
  const float* wT = nullptr; // per-tap transpose [T][Cc][R] (built lazily, cached per sub-step)
  Param* param = nullptr;    // set when the weights ARE a variable (enables the persistent transpose cache)
  SNWeight* sn = nullptr;    // set when the weights are a spectral-norm W/sigma (persistent derived copies live there)
  unsigned short* wb = nullptr;   // bf16 copy, same layout      (compute_dtype bf16; lazily, per sub-step)
  unsigned short* wbT = nullptr;  // bf16 copy, per-tap transposed
  unsigned short* wl = nullptr;   // three bf16 limb planes, same layout   (compute_dtype f32x3; lazily, per sub-step)
  unsigned short* wlT = nullptr;  // three bf16 limb planes, per-tap transposed
  int T = 1, R = 0, Cc = 0;
  std::string name;          // variable name (profiler labels)
  // persistent derived copies owned by somebody else (the gate-filter parts of an SNWeight): per-tap transpose + its valid
  // flag, limb-plane caches [as stored, transposed]
  float* extT = nullptr;
  bool* extT_valid = nullptr;
  LimbCache* extL = nullptr;
};

// persistent per-weight spectral-norm buffers (classifier)
struct SNWeight {
  Param* W = nullptr;
  Param* u = nullptr;
  float *Wbar = nullptr, *WbarT = nullptr, *dWbar = nullptr, *t = nullptr, *sv = nullptr, *dv = nullptr,
        *scal = nullptr;
  bool dWbar_set = false;
  bool wT_valid = false;
  LimbCache wL[2];   // limb planes of Wbar (as stored / transposed), rebuilt when Wbar is
  int K = 0, Co = 0, T = 1;
  // update-gate filters only (conv.py:185-198: the gate convolves concat([prelu(h), x])): Wbar split by input-channel rows into
  // its hidden-state part [9][split_rows][Co] and its image part [9][3][Co] -- persistent copies refreshed with Wbar
  // (sn_prepare), with their own per-tap transposes and limb planes (same life cycle as WbarT / wL)
  int split_rows = 0;
  float* part[2] = {nullptr, nullptr};
  float* partT[2] = {nullptr, nullptr};
  bool partT_valid[2] = {false, false};
  LimbCache partL[2][2];
};

struct ResizeTab {  // separable resampling taps, forward and transposed
  int n_in = 0, n_out = 0, fn = 0, bn = 0;
  int *f_i = nullptr, *b_i = nullptr;    // device
  float *f_w = nullptr, *b_w = nullptr;  // device
};

struct ProfRec {
  const char* name;
  double flops;
  hipEvent_t a, b;
  std::string label;   // "<variable> <fwd|dgrad|wgrad>" of the launching op
  std::string shape;   // "M,N,C,taps,nphase,split"
};

struct Ctx {
  Config cfg;
  hipStream_t stream = nullptr;
  Arena arena;
  Tape tape;
  std::vector<std::unique_ptr<Param>> params;
  std::map<std::string, Param*> by_name;
  NetGroup groups[G_COUNT];
  std::string last_error;

  // per-step state
  std::vector<RedDesc> deferred;   // queued second-stage reductions (ops.cpp: defer_reduction / flush_deferred)
  bool defer_on = true;            // EG_DEFER_REDUCE=0: every reduction launches on the spot
  std::deque<Tensor> tensors;
  std::deque<WRef> wrefs;
  std::map<const void*, WRef*> wref_cache;
  std::deque<bool> flags;   // stable storage for ad-hoc gradient-present flags
  bool sn_prepared = false;   // tape entry for the SN backward pushed in this sub-step
  bool sn_valid = false;      // Wbar buffers match the current D2 weights (persists across sub-steps)
  bool param_grad_enabled = true;
  // set by a caller right before a conv2d / deconv2d whose ONLY reader is in_act: a forward-only K-split launch then
  // leaves its slabs for the InstanceNorm load pass (consumed, i.e. reset, by that convolution)
  // FEED_FLAT / FEED_MINMAX: the same hand-over to a flat element-wise reader (prelu, pool2_add's full-resolution operand)
  // and to the min-max gate (kernels.h: SlabSrc)
  int feeds_in = 0;
  // bf16 mode, set by a caller right before conv2d(..., post_prelu): the PReLU output is read by plane-fed contractions only,
  // so it is stored as its bf16 plane alone (Tensor::d_plane_only; consumed by that convolution)
  bool prelu_plane_only = false;
  bool conv_plane_only_next = false;   // the next conv2d stores its fused-activation output as the bf16 plane alone where the packed-K kernel writes one (ops.cpp)
  T conv_out_into = nullptr;   // conv_into(): the next conv2d writes its result here and records nothing on the tape
  // conv2d_pool2_add(): the pooled outputs the next conv2d offers to its kernel's epilogue, and whether the kernel took them
  struct PoolEpi {
    T b = nullptr, out = nullptr, out2 = nullptr;
    Param* leak = nullptr;
    unsigned short* plane2 = nullptr;
    bool done = false;
  } pool_epi;
  bool conv_wrote_plane = false;   //   ... and reports here whether its kernel stored the output's bf16 plane
  // inference only (set inside infer()): a residual block's 1x1 shortcut convolution runs on this lane beside the block's
  // two 3x3 convolutions on the call's main lane -- the encoder trunk is ONE chain of small launches otherwise (-1: one chain)
  int infer_side = -1;
  float* scratch = nullptr;  // persistent reduction scratch
  size_t scratch_floats = 0;

  // spectral norm
  std::vector<SNWeight> sn;
  std::vector<LimbCache*> limb_caches;   // every persistent limb-plane cache built so far (ops.cpp: limb_planes)
  std::map<Param*, int> sn_index;
  SNDesc* sn_table = nullptr;
  int sn_maxK = 0, sn_maxCo = 0;

  std::map<std::pair<int, int>, ResizeTab> resize_tabs;

  // resident batch (device)
  float *d_images = nullptr, *d_z = nullptr, *d_alpha = nullptr, *d_eps = nullptr;
  float* d_losses = nullptr;  // [64]
  float* d_eval = nullptr;    // eval_losses: [16 losses | 6 x B discriminator logits | 3 x B alpha | 4 eps]
  int* d_classes = nullptr;
  long long step_count = 0;
  // pipelined feed (capi.cpp: eg_train_batch / eg_losses_wait): two pinned host slots for the batch and for the losses
  float* feed_host[2] = {nullptr, nullptr};
  float* loss_host[2] = {nullptr, nullptr};
  hipEvent_t feed_copied[2] = {nullptr, nullptr}, loss_ready[2] = {nullptr, nullptr};
  long long feed_count = 0;
  // generator outputs of the D-phase (sub-steps 1-3 share G1(z), G2(z): G is frozen until sub-step 5)
  float* d_gen_cache[2] = {nullptr, nullptr};
  float* d_gen_bn[2] = {nullptr, nullptr};   // batch statistics of that forward, for the moving-average updates
  bool gen_cache_valid[2] = {false, false};
  // E_norm / G_norm / D_norm = 'batch': statistics of the generators' block norms of the cached forward (moving-average updates of
  // the evaluations that reuse it), the last generator forward's block statistics, and how often a block norm evaluated now
  // advances its moving statistics (the encoder's and the discriminators': once per evaluation of an applied sub-step)
  float* d_gen_bn_blk[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
  T gen_blk_stats[3] = {nullptr, nullptr, nullptr};
  int bn_times_block = 0;

  // Lanes: sub-steps without data dependencies run on separate streams (own arena and reduction scratch each) so
  // that the HBM-bound kernels of one overlap the MFMA-bound kernels of another; `stream` / `arena` / `scratch`
  // above always belong to the ACTIVE lane (use_lane swaps them), lanes[i].arena parks lane i's arena meanwhile.
  struct Lane {
    Arena arena;
    hipStream_t stream = nullptr;
    float* scratch = nullptr;
    hipEvent_t done = nullptr;
    // "leaf" stream of the lane: filter / bias gradients are leaves of the backward graph (nothing downstream in the
    // sweep reads them), so they run here, beside the lane's dependent chain dgrad -> norm / activation backward -> dgrad
    // whose HBM-bound kernels they overlap (ops.cpp: LeafScope); joined where their results are read (flush_deferred)
    hipStream_t leaf = nullptr;
    hipEvent_t leaf_in = nullptr, leaf_out = nullptr;
  };
  static constexpr int kTrainLanes = 4;   // the lanes train_step schedules on: created in build_context, in the order the hardware-queue placement assumes
  static constexpr int kLanes = 8;        // + the extra inference slots' lanes (main stream, arena, scratch; no leaf stream), created at first use
  Lane lanes[kLanes];
  int cur_lane = 0;
  bool lanes_on = true;
  int lanes_mask = 15;
  bool leaf_on = true;      // EG_LANES bit 3
  WRef* alt_w = nullptr;    // ops.cpp attach_bf16 -> run_conv: the filter of the launch being prepared (64 .. 127 output channels,
  bool alt_transposed = false;   //   bf16 mode), should run_conv decide for the plane-fed kernel and need its K-step-major plane
  bool leaf_dirty = false;  // the active lane's leaf stream holds work the main stream has not ordered itself behind
  std::vector<hipStream_t> placeholder_streams;   // model.cpp: stream -> hardware queue placement
  hipEvent_t ev_fork = nullptr, ev_gen = nullptr, ev_g1fwd = nullptr, ev_branch = nullptr, ev_cls = nullptr;

  // inference staging buffers (host <-> device hand-over of eg_infer), grown on demand and kept
  // Consecutive eg_infer_run calls on a resident batch are independent of one another: they rotate over kInferSlots slots
  // (own lanes = streams + arenas, own output buffers), so several calls are in flight at once and their small launches
  // (4x4 .. 64x64 maps: most fill a fraction of the chip) share the device (model.cpp: infer, capi.cpp: eg_infer_run)
  static constexpr int kInferSlots = 8;
  float *inf_in = nullptr, *inf_eps = nullptr;
  float *inf_edge[kInferSlots] = {}, *inf_image[kInferSlots] = {};
  int* inf_cls = nullptr;
  size_t inf_cap = 0;   // images the buffers hold
  int inf_nb = 0;       // images currently resident (eg_infer_upload)
  long long inf_runs = 0;   // eg_infer_run calls since the context was created (slot = inf_runs mod the slots in use)
  int inf_last_slot = 0;    // slot of the latest run: what eg_infer_download returns
  int inf_slots = 4;        // calls in flight x lanes per call: 8 / 4 = eight / four single-chain calls, 2 = two calls of two lanes each, 1 = one call of two lanes (EG_INFER_SLOTS)
  hipEvent_t ev_infer[kLanes] = {};   // fork event of the call whose main lane is i
  bool inf_inflight = false;   // eg_infer_run calls may still be running on lanes other than 0: every OTHER entry point drains them first (api_enter)
  bool inf_entering = false;   // ... set by eg_infer_run around its own api_enter
  bool inf_warm = false;       // the weights' lazily built derived copies (transposes, bf16 planes) are all in place: set by the first
                               // completed run, cleared whenever a variable changes -- until then a run is followed by a device sync

  // profiling
  bool prof_on = false;
  std::vector<ProfRec> prof;
  std::vector<hipEvent_t> ev_pool;
  std::string prof_report;
  std::string cur_label;   // set by the op that launches next (profiling only)

  // data parallel (RCCL, loaded lazily)
  void* nccl_lib = nullptr;
  void* nccl_comm = nullptr;
  int rank = 0, nranks = 1;
  // gradient exchange + optimizer of a group run on a side stream so they overlap the next sub-step's compute
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_ready[G_COUNT] = {}, ev_applied[G_COUNT] = {};
  bool pending[G_COUNT] = {};
  float* comm_scal = nullptr;   // device staging of comm_reduce_scalars
  long long comm_calls = 0, comm_bytes = 0;   // gradient all-reduces enqueued / their payload (comm_info)
  int comm_cus = 0;             // EG_COMM_CUS: compute units the communication stream is confined to (0: no mask)
  hipStream_t apply_stream = nullptr;   // ... then the optimizer behind an exchange runs here, unmasked (model.cpp: comm_apply_stream)
  bool comm_timing = false;     // eg_comm_timing: event pairs around every gradient all-reduce
  struct CommRec {
    int group = 0;
    long long bytes = 0;
    hipEvent_t a = nullptr, b = nullptr;
  };
  std::deque<CommRec> comm_recs;

  ~Ctx();
};

// ---- registry (model.cpp) ----------------------------------------------------------------------
struct VarInfo {
  std::string name;
  std::vector<int> shape;
  bool trainable, dead;
  int init_kind;
  float init_param;
  int group;
};
std::vector<VarInfo> enumerate_variables(const Config& cfg);
void build_context(Ctx& c);
Param* P(Ctx& c, const std::string& name);

// ---- tape ops (ops.cpp) --------------------------------------------------------------------------
T new_tensor(Ctx& c, int n, int h, int w, int ch, bool needs_grad, bool alloc = true);
T wrap_tensor(Ctx& c, float* data, int n, int h, int w, int ch);
float* grad_acc(Ctx& c, T t, int* acc);   // returns t->g; *acc = 1 if it already holds a value
void reset_step(Ctx& c);                  // arena + tape + tensor list (start of a sub-step)
void flush_deferred(Ctx& c);              // runs the queued second-stage reductions on the current stream
void api_enter(Ctx& c);                   // model.cpp: entry state of every enqueueing C-ABI call (recovers from a failed one)
void leaf_join(Ctx& c);                   // the lane's main stream waits for its leaf stream (ops.cpp: LeafScope)
void reset_grads(Ctx& c);                 // forget every activation / weight gradient (keeps buffers)
void set_want_grad(Ctx& c, int group, bool on);

WRef* wref_param(Ctx& c, Param* p, int T, int R, int Cc);
WRef* wref_sn(Ctx& c, Param* p);
const float* wref_transposed(Ctx& c, WRef* w);

enum PadMode { PAD_SAME = 0, PAD_REFLECT = 1 };
// y = conv(x, w) (+ bias); with accumulate_into the result is added to an existing output (split-input convs)
// post_prelu: returns prelu(conv(x) + bias, post_prelu) -- the convolution's epilogue stores it as a second output where its
// kernel can, a K-split launch hands its slabs to the PReLU pass, everything else runs prelu() behind the convolution
T conv2d(Ctx& c, T x, WRef* w, Param* bias, int k, int stride, int pad, T accumulate_into = nullptr,
         int fused_act = 0 /* ACT_RELU / ACT_LRELU applied by the conv where its kernel can (else by act()) */,
         Param* post_prelu = nullptr);
// fused_act == ACT_TANH: the caller applies tanh to the result next; a forward-only 3-channel layer (the generators' last) then
// stores tanh(y) itself where its kernel can (returned tensor: fused_act == ACT_TANH).  y_into: the result is written there
T deconv2d(Ctx& c, T x, WRef* w /*[25][Cout][Cin]*/, Param* bias, int cout, int fused_act = 0, float* y_into = nullptr);
T linear(Ctx& c, T x, WRef* w /*[1][in][out]*/, Param* bias);
T linear1(Ctx& c, T x, Param* w /*[K][1]*/, Param* bias /*nullable*/);
T reshape(Ctx& c, T src, int n, int h, int w, int ch);
// ---- explicit (tape-free) pieces for schedules the linear tape cannot express: the stacked discriminator sub-step (model.cpp) ----
// images [n0, n0 + n) of a tensor as a tensor of their own: data, gradient and bf16 planes alias the parent's (the parent's
// gradient buffer is allocated here if needed); no tape entry, g_set is the view's own
void tensor_need_f32(Ctx& c, T t);   // widen a plane-only tensor (Tensor::d_plane_only) for a reader of its fp32 values
T view_rows(Ctx& c, T parent, int n0, int n);
// y = conv(x, w) (SAME) (+ ReLU / LeakyReLU where the thin kernels fuse it), forward only, into y.  Returns bit 0: the activation
// was applied, bit 1: the kernel also stored y's bf16 plane rows (bf16 mode, y->d_bf given)
int conv_into(Ctx& c, T x, WRef* w, T y, int k, int stride, int fused_act = 0);
// filter gradient (on the lane's leaf stream) / input gradient of y = conv(x, w) (SAME) from y->g; x->g first-written or added to
void conv_backward_w(Ctx& c, T x, T y, WRef* w, int k, int stride);
void conv_backward_x(Ctx& c, T x, T y, WRef* w, int k, int stride);
// N == 1 head (linear1): dW (+)= sum over `rows` rows of x[r] dy[r]; the bias gradient sums the first bias_rows of them only (on
// the lane's leaf stream)
void linear1_backward_w(Ctx& c, const float* x, const float* dy, Param* w, Param* bias, int bias_rows, int rows, int K);
WRef* wref_raw(Ctx& c, const float* w_, float* g, bool* g_set, bool want, int T_, int R, int Cc,
               const std::string& name = std::string());
T in_act(Ctx& c, T x, int act, T* stats_out = nullptr, bool plane_only = false);
T in_act_tangent(Ctx& c, T tx, T x_prim, T out_prim, T stats_prim, int act);
T act(Ctx& c, T x, int a);
T act_tangent(Ctx& c, T tx, T out_prim, int a);
float* in_chunk_scratch(Ctx& c, int NB, int HW, int C);   // ops.cpp: scratch of the chunked InstanceNorm kernels (large maps), or null
T bn_relu(Ctx& c, T x, Param* gamma, Param* beta, Param* mm, Param* mv, int update_times, T* stats_out = nullptr);
T bn_act(Ctx& c, T x, Param* gamma, Param* beta, Param* mm, Param* mv, int act, int update_times, T* stats_out = nullptr);
T bn_act_tangent(Ctx& c, T tx, T x_prim, T out_prim, T stats_prim, Param* gamma, int act);
T add(Ctx& c, T a, T b, bool relu);
T pool2(Ctx& c, T x);
T add_relu_pool2(Ctx& c, T a, T b);   // pool2(relu(a + b)): one pass in forward-only graphs
// pool2(a + b) in one pass (the MRU unit's `mean_pool(proj + h_new)`, conv.py:239-243); a and b must have no other
// consumer: their gradients are then ONE shared buffer (both equal the up-sampled output gradient / 4)
T add_pool2(Ctx& c, T a, T b);
// pool2(a) + b, b at the pooled resolution and with no other consumer (its gradient IS the output's gradient buffer)
// post_prelu / prelu_out: also returns prelu(result, post_prelu) from the same pass (conv.py:185: the next unit's prelu(h))
// plane_only (bf16 mode): that PReLU output is read by plane-fed contractions only and stored as its bf16 plane alone
T pool2_add(Ctx& c, T a, T b, Param* post_prelu = nullptr, T* prelu_out = nullptr, bool plane_only = false);
// pool2_add(conv2d(x, w) + bias, b, ...) with the pooling done in the convolution's epilogue where its kernel can (the
// full-resolution convolution output is then never stored; conv.py:229-243); otherwise the two ops as usual
T conv2d_pool2_add(Ctx& c, T x, WRef* w, Param* bias, int k, T b, Param* post_prelu = nullptr, T* prelu_out = nullptr,
                   bool plane_only = false);
T gmean(Ctx& c, T x, bool pre_relu);
T resize(Ctx& c, T x, int size);
T concat_w(Ctx& c, T a, T b);
T slice_w(Ctx& c, T x, int w0, int w1);   // no gradient (inputs only)
T prelu(Ctx& c, T x, Param* leak, float leak_const);
T minmax(Ctx& c, T x, float pre_leak = 0.f);   // pre_leak != 0: minmax(lrelu(x)) in one pass each way
T gate(Ctx& c, T h, T rg, T img);
T gate_prelu(Ctx& c, T h, T rg, T img, Param* leak, bool plane_only = false);   // prelu(h + rg * img, leak) in one pass each way
T reparam(Ctx& c, T mu, T ls, const float* eps_dev);
void sn_prepare(Ctx& c);                  // W/sigma for every live SN weight + tape entry for its backward
void seed_grad(Ctx& c, T t);              // marks t->g as holding a value (after a loss kernel wrote it)

struct ProfScope {
  Ctx& c;
  int idx = -1;
  explicit ProfScope(Ctx& c_);
  void done(const char* name, double flops, const std::string& shape = std::string());
};

// ---- networks and sub-steps (model.cpp) --------------------------------------------------------
struct DiscSave {
  T pre[4] = {nullptr, nullptr, nullptr, nullptr};
  T out[4] = {nullptr, nullptr, nullptr, nullptr};
  T stats[4] = {nullptr, nullptr, nullptr, nullptr};
};
// out_into: the network's output is written there (inference: the caller's output buffer, no copy behind the network)
T generator_forward(Ctx& c, int group, T zin, int bn_times, T* bn_stats_out = nullptr, float* out_into = nullptr);
T disc_forward(Ctx& c, int group, T x, DiscSave* save, const DiscSave* tangent_of);
// z1_out: a forward-only graph may also return z for a SECOND eps draw, eps_dev[1] (inference: the E -> G2 run), from the same
// launch; null where the head ran as separate ops
T encoder_forward(Ctx& c, T x, const float* eps_dev, T* mu_out = nullptr, T* ls_out = nullptr, T* z1_out = nullptr);
T classifier_forward(Ctx& c, T x);

enum LossSlot {
  L_JOINT_D = 0, L_JOINT_GP, L_IMAGE_D, L_IMAGE_GP, L_EDGE_D, L_EDGE_GP, L_CLS_D, L_JOINT_G, L_EDGE_G, L_IMAGE_G,
  L_CLS_G, L_ZL, L_COUNT
};
// k in 1..7 (edgegan.py:109-124 order); apply=false computes gradients only (parity tests)
// Adjustments train_step makes when it runs independent sub-steps concurrently (all null / -1 = the plain
// sequential meaning of the sub-step).
struct SubOpts {
  int bn_times_g1 = -1, bn_times_g2 = -1;   // BatchNorm moving-statistics updates to apply for G1 / G2 in this sub-step
  hipEvent_t record_after_gen = nullptr;      // recorded once the shared G1(z), G2(z) are available (sub-step 1)
  hipEvent_t record_after_g1_fwd = nullptr;   // recorded after G1's forward has been enqueued (sub-step 6)
  hipEvent_t wait_before_apply = nullptr;     // the optimizer of this sub-step may not start before this event
  // sub-step 5 right behind the concurrent sub-steps 1-4: instead of waiting for all of them up front, wait for a
  // network's update (event of the lane that trained it) only where that network is first used
  const hipEvent_t* wait_for_group = nullptr;  // [G_COUNT], null entries = no wait
  // sub-steps 5 / 7: the four critics' forward + input-gradient passes are independent given G1(z), G2(z); run each on the
  // lane that trained that network (D: 0, D_patch2: 1, D_patch3: 2, classifier: 3), join before the generators' backward
  bool branch_lanes = false;
};
void run_substep(Ctx& c, int k, bool apply, const SubOpts* opts = nullptr);
void train_step(Ctx& c);
struct EvalResult {   // device pointers valid until the next enqueueing call
  const float *edge = nullptr, *image = nullptr, *resized[4] = {nullptr, nullptr, nullptr, nullptr};
};
void eval_losses(Ctx& c, const float* alpha_dev, const float* eps_dev, int bn_updates, EvalResult* r);
void infer(Ctx& c, const float* d_images, const int* d_classes, int NB, const float* d_eps2, float* d_edge,
           float* d_image, int main_lane = 0, int side_lane = 1);

// ---- data parallel (comm.cpp) -----------------------------------------------------------------
void comm_unique_id(Ctx& c, char* out128);
void comm_init(Ctx& c, const char* id128, int rank, int nranks);
void comm_allreduce(Ctx& c, float* buf, size_t n, hipStream_t stream, int group);
int comm_timing_read(Ctx& c, float* ms, int* group, long long* bytes, int cap);
void comm_cu_mask(int ncu, uint32_t out[8]);   // EG_COMM_CUS (comm.cpp)
int comm_reserved_cus();
void comm_reduce_scalars(Ctx& c, float* inout, int n, int op);
void use_lane(Ctx& c, int lane);                // model.cpp
void comm_wait_groups(Ctx& c, unsigned mask);   // main stream waits for the pending updates of these groups
void comm_wait_all(Ctx& c);
void comm_info(Ctx& c, int* version, char* libpath, int cap, long long* calls, long long* bytes);
void comm_destroy(Ctx& c);

}  // namespace eg
