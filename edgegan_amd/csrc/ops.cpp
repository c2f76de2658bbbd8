// Differentiable ops of the EdgeGAN hot path on a reverse tape.  Each op launches gfx950 kernels
// for its forward and records a closure that launches the kernels of its backward.
// Reference semantics are cited per op; TF-1.14 op conventions follow SURVEY.md Appendix B.
#include <cmath>
#include <cstring>

#include "engine.h"

namespace eg {

// ---------------------------------------------------------------------------------------------
// profiling scope: HIP events on the engine's own stream around one launch
// ---------------------------------------------------------------------------------------------
static hipEvent_t get_event(Ctx& c) {
  if (!c.ev_pool.empty()) {
    hipEvent_t e = c.ev_pool.back();
    c.ev_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  EG_HIP(hipEventCreate(&e));
  return e;
}
ProfScope::ProfScope(Ctx& c_) : c(c_) {
  if (c.prof_on) {
    ProfRec r{"", 0.0, get_event(c), get_event(c), std::string(), std::string()};
    EG_HIP(hipEventRecord(r.a, c.stream));
    c.prof.push_back(r);
    idx = (int)c.prof.size() - 1;
  }
}
void ProfScope::done(const char* name, double flops, const std::string& shape) {
  if (idx >= 0) {
    c.prof[idx].name = name;
    c.prof[idx].flops = flops;
    c.prof[idx].label = c.cur_label;
    c.prof[idx].shape = shape;
    EG_HIP(hipEventRecord(c.prof[idx].b, c.stream));
  }
}

// ---------------------------------------------------------------------------------------------
// tensors / step state
// ---------------------------------------------------------------------------------------------
T new_tensor(Ctx& c, int n, int h, int w, int ch, bool needs_grad, bool alloc) {
  c.tensors.emplace_back();
  T t = &c.tensors.back();
  t->n = n;
  t->h = h;
  t->w = w;
  t->c = ch;
  t->needs_grad = needs_grad;
  if (alloc) t->d = c.arena.allocf(t->numel());
  return t;
}
T wrap_tensor(Ctx& c, float* data, int n, int h, int w, int ch) {
  T t = new_tensor(c, n, h, w, ch, false, false);
  t->d = data;
  return t;
}
// Second-stage reductions (kernels.h: RedBatch) are queued and run one launch per batch.  Everything that READS a
// gradient a queued reduction writes calls flush_deferred first: the end of a backward pass (Tape::after), the gate
// filter's column copy, the spectral-norm backward, and a second reduction into the same buffer (summation order kept).
static void run_deferred(Ctx& c, hipStream_t s) {
  if (c.deferred.empty()) return;
  RedBatch b{};
  b.count = (int)c.deferred.size();
  unsigned at = 0;
  for (int i = 0; i < b.count; ++i) {
    b.d[i] = c.deferred[i];
    b.start[i] = at;
    at += red_desc_blocks(b.d[i]);
  }
  b.start[b.count] = at;
  launch_reduce_batch(b, s);
  c.deferred.clear();
}
// Leaf stream of the active lane (engine.h: Lane::leaf): filter and bias gradients are enqueued there, behind an event
// that marks the moment their inputs (the layer's output gradient) are final on the lane's main stream.  The queued
// reductions follow them on the leaf stream; flush_deferred -- called exactly where a parameter gradient is read on the
// main stream -- orders the main stream behind everything the leaf stream holds.  Same kernels, same per-buffer order
// (all writers of one filter gradient are on the one leaf stream): bitwise the results of the one-stream schedule.
// one bit per stream the context enqueues on: lanes' main streams 0 .., their leaf streams 16 ..
static unsigned stream_bit(Ctx& c) { return 1u << (c.stream == c.lanes[c.cur_lane].stream ? c.cur_lane : 16 + c.cur_lane); }
static bool leaf_active(Ctx& c) { return c.leaf_on && c.lanes_on && c.defer_on && !c.prof_on && c.lanes[c.cur_lane].leaf; }
struct LeafScope {
  Ctx& c;
  hipStream_t main;
  bool on;
  explicit LeafScope(Ctx& c_) : c(c_), main(c_.stream), on(leaf_active(c_)) {
    if (!on) return;
    Ctx::Lane& l = c.lanes[c.cur_lane];
    EG_HIP(hipEventRecord(l.leaf_in, main));
    EG_HIP(hipStreamWaitEvent(l.leaf, l.leaf_in, 0));
    c.stream = l.leaf;
    c.leaf_dirty = true;
  }
  ~LeafScope() {
    if (on) c.stream = main;
  }
};
// The main stream orders itself behind everything its lane's leaf stream holds (no-op when there is nothing).
void leaf_join(Ctx& c) {
  if (!c.leaf_dirty || !leaf_active(c) || c.stream != c.lanes[c.cur_lane].stream) return;
  Ctx::Lane& l = c.lanes[c.cur_lane];
  EG_HIP(hipEventRecord(l.leaf_out, l.leaf));
  EG_HIP(hipStreamWaitEvent(c.stream, l.leaf_out, 0));
  c.leaf_dirty = false;
}
void flush_deferred(Ctx& c) {
  if (leaf_active(c) && c.stream == c.lanes[c.cur_lane].stream) {
    if (!c.deferred.empty()) {
      run_deferred(c, c.lanes[c.cur_lane].leaf);
      c.leaf_dirty = true;
    }
    leaf_join(c);
    return;
  }
  run_deferred(c, c.stream);
}
static void defer_reduction(Ctx& c, const RedDesc& d) {
  for (const RedDesc& q : c.deferred)
    if (q.dst == d.dst) {
      run_deferred(c, c.stream);
      break;
    }
  c.deferred.push_back(d);
  if ((int)c.deferred.size() == kRedBatch) run_deferred(c, c.stream);
}

// SlabSrc views of a tensor's value / gradient (kernels.h), and the stand-alone sum for readers that want a plain tensor
static SlabSrc data_src(T t) {
  if (t->nslab > 1) return SlabSrc{t->slabs, t->nslab, t->c, (unsigned long long)t->numel(), t->slab_bias, t->slab_base};
  return plain_src(t->d);
}
static SlabSrc grad_src(T t) {
  if (t->g_nslab > 1) return SlabSrc{t->g_slabs, t->g_nslab, t->c, (unsigned long long)t->numel(), nullptr, nullptr};
  return plain_src(t->g);
}
static void materialize_grad(Ctx& c, T t) {
  if (t->g_nslab <= 1) return;
  launch_conv_split_reduce(t->g_slabs, t->g, nullptr, t->numel(), t->c, t->c, t->g_nslab, 0, c.stream);
  t->g_nslab = 0;
  t->g_slabs = nullptr;
}
// Tensor::g_up: write the gradient out after all (a reader without the half-resolution form)
static void materialize_up(Ctx& c, T t) {
  if (!t->g_up) return;
  if (!t->g) t->g = c.arena.allocf(t->numel());
  launch_pool2_bwd(t->g_up->g, t->g, 0, t->n, t->h, t->w, t->c, c.stream, nullptr);
  t->g_up = nullptr;
}
// Tensor::g_plane_only.  plane_only_acc: the `accumulate` argument of a gradient writer that stores `plane` -- with GRAD_PLANE_ONLY
// where the fp32 gradient of t need not exist; grad_f32_view: the fp32 gradient for a reader without a plane-fed form (its own
// widened copy on its own stream: the filter gradient runs on the leaf stream, the input gradient on the main one).
static int plane_only_acc(Ctx& c, T t, int acc, const unsigned short* plane) {
  static const bool on = dbg_env("EG_NO_PLANE_ONLY_GRAD") == nullptr;
  if (!on || !c.cfg.bf16 || !t->g_plane_ok || acc || !plane) return acc;
  t->g_plane_only = true;
  return GRAD_PLANE_ONLY;
}
static const float* grad_f32_view(Ctx& c, T t) {
  if (!t->g_plane_only) return t->g;
  float* full = c.arena.allocf(t->numel());
  launch_cvt_from_bf16(t->g_bf, full, t->numel(), c.stream);
  if (dbg_env("EG_TRACE_CVT")) fprintf(stderr, "[widen] grad view n=%d h=%d w=%d c=%d lane=%d\n", t->n, t->h, t->w, t->c, c.cur_lane);
  return full;
}
// Column partials of the gradient a backward kernel is about to write as the FIRST writer (acc == 0) of tensor t, if t's producer
// asked for them (a convolution with a trainable bias: Tensor::want_colpart) and the kernel form has `rows` > 0 of them
static float* colpart_for(Ctx& c, T t, int acc, int rows) {
  static const bool on = dbg_env("EG_NO_COLPART") == nullptr;
  if (!on || acc || !t->want_colpart || rows <= 0 || !c.param_grad_enabled || !c.defer_on) return nullptr;
  return c.arena.allocf((size_t)rows * t->c);
}
static void colpart_done(T t, float* part, int rows) {   // (after grad_acc, which cleared any older table)
  if (!part) return;
  t->g_colpart = part;
  t->g_colpart_rows = rows;
}
// the bias gradient of a convolution from what its output's gradient carries: the first writer's column partials, a smaller
// tensor with the same column sums (pool2_add), or a pass over the gradient itself
static void bias_grad(Ctx& c, Param* b, const float* dy, size_t rows, int C);
static void bias_grad_of(Ctx& c, Param* b, T y) {
  if (y->g_colpart && c.defer_on) {
    defer_reduction(c, RedDesc{y->g_colpart, b->g, 1, 0, y->c, 0, 0, 0, y->g_colpart_rows, b->g_set ? 1 : 0});
    b->g_set = true;
  } else if (y->g_colsum_src) {
    bias_grad(c, b, y->g_colsum_src, y->g_colsum_rows, y->c);   // (pool2_add: a quarter of the rows)
  } else {
    materialize_up(c, y);
    bias_grad(c, b, grad_f32_view(c, y), y->pixels(), y->c);
  }
}

// flat element-wise readers (prelu, gate_prelu, pool2 in their float4 forms) take any [..][C] tensor with C % 4 == 0
static bool flat_slab_ok(T t) { return t->c % 4 == 0 && t->numel() < (1ull << 31); }

float* grad_acc(Ctx& c, T t, int* acc) {
  if (!t->g) t->g = c.arena.allocf(t->numel());
  EG_CHECK(!t->g_up, "a second gradient contribution reached a tensor whose gradient is held as its pooled form");
  if (t->g_set && t->g_plane_only) {   // (a second writer adds to the fp32 value)
    launch_cvt_from_bf16(t->g_bf, t->g, t->numel(), c.stream);
    if (dbg_env("EG_TRACE_CVT")) fprintf(stderr, "[widen] grad second writer n=%d h=%d w=%d c=%d lane=%d\n", t->n, t->h, t->w, t->c, c.cur_lane);
  }
  t->g_plane_only = false;
  if (t->g_set) materialize_grad(c, t);   // (the first writer left K-split slabs: a second one adds to their sum)
  // ConvArgs::epi_act: the FIRST writer applied the fused activation's derivative in its epilogue and the producer will
  // skip its in-place pass -- exact only if that writer stays the ONLY one.  A second consumer accumulating a raw
  // dL/d(out) into the same buffer would go unmasked: refuse loudly (no graph of this engine does it today).
  EG_CHECK(!(t->g_set && t->g_act_done), "a second gradient contribution reached a tensor whose activation backward was fused into the first writer's epilogue");
  *acc = t->g_set ? 1 : 0;
  t->g_set = true;
  t->g_bf = nullptr;   // whoever asks is about to write g: a plane of its previous value is stale (the writer may set a new one)
  t->g_colsum_src = nullptr;
  t->g_colpart = nullptr;
  return t->g;
}
void seed_grad(Ctx& c, T t) {
  if (!t->g) t->g = c.arena.allocf(t->numel());
  t->g_set = true;
}
void reset_grads(Ctx& c) {
  for (auto& t : c.tensors) {
    t.g_set = t.g_act_done = false;
    t.g_bf = nullptr;   // (the plane of a gradient is as short-lived as the gradient)
    t.g_nslab = 0;
    t.g_slabs = nullptr;
    t.g_colsum_src = nullptr;
    t.g_colpart = nullptr;
    t.g_up = nullptr;
    t.g_plane_only = false;
  }
  for (auto& p : c.params) p->g_set = false;
  for (auto& s : c.sn) s.dWbar_set = false;
}
void reset_step(Ctx& c) {
  flush_deferred(c);   // (empty unless a caller skipped its backward pass)
  c.arena.reset();
  c.tape.clear();
  c.tensors.clear();
  c.wrefs.clear();
  c.wref_cache.clear();
  c.flags.clear();
  c.sn_prepared = false;
  c.param_grad_enabled = true;
  c.alt_w = nullptr;   // (points into c.wrefs)
  for (auto& p : c.params) p->g_set = false;
  for (auto& s : c.sn) s.dWbar_set = false;
}
void set_want_grad(Ctx& c, int group, bool on) {
  for (Param* p : c.groups[group].params) p->want_grad = on && p->trainable && !p->dead;
}

WRef* wref_param(Ctx& c, Param* p, int T_, int R, int Cc) {
  auto it = c.wref_cache.find(p);
  if (it != c.wref_cache.end()) return it->second;
  EG_CHECK((size_t)T_ * R * Cc == p->numel, "wref shape mismatch for %s", p->name.c_str());
  c.wrefs.emplace_back();
  WRef* w = &c.wrefs.back();
  w->w = p->w;
  w->g = p->g;
  w->g_set = &p->g_set;
  w->want_grad = p->want_grad;
  w->T = T_;
  w->R = R;
  w->Cc = Cc;
  w->param = p;
  w->name = p->name;
  c.wref_cache[p] = w;
  return w;
}
WRef* wref_sn(Ctx& c, Param* p) {
  auto it = c.wref_cache.find(p);
  if (it != c.wref_cache.end()) return it->second;
  SNWeight& s = c.sn[c.sn_index.at(p)];
  c.wrefs.emplace_back();
  WRef* w = &c.wrefs.back();
  w->w = s.Wbar;
  w->sn = &s;
  w->g = s.dWbar;
  w->g_set = &s.dWbar_set;
  w->want_grad = p->want_grad;
  w->T = s.T;
  w->R = s.K / s.T;
  w->Cc = s.Co;
  if (s.wT_valid) w->wT = s.WbarT;
  w->name = p->name;
  c.wref_cache[p] = w;
  return w;
}
// an ad-hoc dense weight (e.g. a channel slice of an SN filter) living in the arena
WRef* wref_raw(Ctx& c, const float* w_, float* g, bool* g_set, bool want, int T_, int R, int Cc,
               const std::string& name) {
  c.wrefs.emplace_back();
  WRef* w = &c.wrefs.back();
  w->w = w_;
  w->g = g;
  w->g_set = g_set;
  w->want_grad = want;
  w->T = T_;
  w->R = R;
  w->Cc = Cc;
  w->name = name;
  return w;
}
const float* wref_transposed(Ctx& c, WRef* w) {
  if (w->wT) return w->wT;
  if (w->extT) {
    if (!*w->extT_valid) {
      launch_tap_transpose(w->w, w->extT, w->T, w->R, w->Cc, nullptr, c.stream);
      *w->extT_valid = true;
    }
    w->wT = w->extT;
    return w->wT;
  }
  if (w->param && w->param->cache_T) {
    Param* p = w->param;
    if (!p->wT) EG_HIP(hipMalloc(&p->wT, p->numel * sizeof(float)));
    if (!p->wT_valid) {
      launch_tap_transpose(w->w, p->wT, w->T, w->R, w->Cc, nullptr, c.stream);
      p->wT_valid = true;
    }
    w->wT = p->wT;
    return w->wT;
  }
  float* buf = nullptr;
  // SN weights keep their transpose in a persistent buffer
  for (auto& s : c.sn)
    if (s.Wbar == w->w) {
      buf = s.WbarT;
      s.wT_valid = true;
      break;
    }
  if (!buf) buf = c.arena.allocf((size_t)w->T * w->R * w->Cc);
  launch_tap_transpose(w->w, buf, w->T, w->R, w->Cc, nullptr, c.stream);
  w->wT = buf;
  return buf;
}

static void run_conv(Ctx& c, ConvArgs& a, T xsrc = nullptr, bool xgrad = false, float* x_full = nullptr);

// Phase-merged 3x3 filter [9][C][12] of a stride-2 filter [k*k][3][C] with 3 image channels (kernels.h:
// launch_merge_phase_filter); kept per variable until the variable changes, like the per-tap transposes.
static const float* merged_filter(Ctx& c, WRef* w, int k, int C) {
  const size_t n = (size_t)9 * C * 12;
  if (w->param && w->param->cache_T) {
    Param* p = w->param;
    if (!p->wM) EG_HIP(hipMalloc(&p->wM, n * sizeof(float)));
    if (!p->wM_valid) {
      launch_merge_phase_filter(w->w, p->wM, k, C, c.stream);
      p->wM_valid = true;
    }
    return p->wM;
  }
  float* buf = c.arena.allocf(n);
  launch_merge_phase_filter(w->w, buf, k, C, c.stream);
  return buf;
}

// One launch for all 4 sub-pixel phases of a 3-image-channel stride-2 layer: a 3x3 stride-1 conv over the coarse grid
// [NB][IH][IW][C] producing 12 channels that the epilogue scatters to the 2x2 output pixels ([NB][2IH][2IW][3]).
// Against 4 phase launches with a 3 -> 16 padded tile this reads the coarse operand 9 instead of 16 times and issues
// 9 x 16 instead of 16 x 16 padded MACs per coarse pixel and channel.
// act == ACT_TANH: the output activation rides in the store where the kernel that runs has that form; returns whether it did
static bool run_merged_phases(Ctx& c, const float* x, int NB, int IH, int IW, int C, const float* wm, const float* bias,
                              float* y, int accumulate, int act = 0) {
  ConvArgs a{};
  a.act = accumulate ? 0 : act;
  a.x = x;
  a.wgt = wm;
  a.bias = bias;
  a.y = y;
  a.NB = NB;
  a.IH = IH;
  a.IW = IW;
  a.C = C;
  a.x_ps = C;
  a.OHf = 2 * IH;
  a.OWf = 2 * IW;
  a.N = 12;
  a.y_ps = 3;
  a.KW = 3;
  a.kst = 1;
  a.S = 1;
  a.D = 1;
  a.os = 1;
  a.w_ts = (long long)C * 12;
  a.ldw = 12;
  a.accumulate = accumulate;
  a.shuf = 1;
  a.nphase = 1;
  a.ph[0] = ConvPhase{IH, IW, 0, 0, 3, 3, 0, 0, -1, -1};
  run_conv(c, a);
  return a.act == ACT_TANH && (a.caps & CONV_CAP_TANH);
}

// compute_dtype bf16: hand the conv a bf16 copy of the filter with the contracted index contiguous.  `transposed`
// selects the per-tap transpose [T][Cc][R] (contracted index = R) or the stored layout [T][R][Cc] (contracted = Cc).
// bf16 limb planes (3: f32x3 mode, 1: the value rounded to bf16) of a filter in the orientation a launch needs.  Network
// variables and spectral-norm weights keep them until the weights change (a discriminator's filters are read in its own
// sub-step, then unchanged in sub-steps 5 and 7); ad-hoc weights (filter slices) get per-sub-step copies in the arena.
static unsigned short* limb_planes(Ctx& c, WRef* w, bool transposed, int nplanes) {
  const size_t n = (size_t)w->T * w->R * w->Cc;
  const int o = transposed ? 1 : 0;
  LimbCache* lc = nullptr;
  if (w->param && w->param->cache_T)
    lc = &w->param->wL[o];
  else if (w->sn)
    lc = &w->sn->wL[o];
  else if (w->extL)
    lc = &w->extL[o];
  if (lc) {
    if (lc->buf && lc->planes != nplanes) {
      (void)hipFree(lc->buf);
      lc->buf = nullptr;
    }
    if (!lc->buf) {
      EG_HIP(hipMalloc(&lc->buf, (size_t)nplanes * n * sizeof(unsigned short)));
      lc->planes = nplanes;
      lc->valid = false;
    }
    if (!lc->ready) EG_HIP(hipEventCreateWithFlags(&lc->ready, hipEventDisableTiming));
    if (!lc->src) {   // first build: remember the source and find the domain (the variable's network group / SN weights)
      lc->src = w->w;
      lc->T = w->T;
      lc->R = w->R;
      lc->Cc = w->Cc;
      lc->o = o;
      lc->domain = G_COUNT;
      if (!w->sn)
        for (int g = 0; g < G_COUNT; ++g)
          if (c.groups[g].w && w->w >= c.groups[g].w && w->w < c.groups[g].w + c.groups[g].n_total) lc->domain = g;
      c.limb_caches.push_back(lc);
    }
    if (!lc->valid) {
      // every stale cache of the domain goes into the same launch: the domain's weights changed together (one RMSProp
      // launch per group / one sn_prepare) and are final until the next update
      static const bool batch_on = dbg_env("EG_NO_SPLIT_BATCH") == nullptr;
      SplitBatch b;
      b.count = 0;
      b.start[0] = 0;
      std::vector<LimbCache*> built;
      auto flush = [&]() {
        launch_split_bf16x3_batch(b, c.stream);
        b.count = 0;
        b.start[0] = 0;
      };
      for (LimbCache* q : c.limb_caches) {
        if (q != lc && (!batch_on || q->valid || !q->buf || q->domain != lc->domain || q->planes != nplanes)) continue;
        SplitDesc& d = b.d[b.count];
        d.in = q->src;
        d.out = q->buf;
        d.plane = (unsigned long long)q->T * q->R * q->Cc;
        d.R = q->R;
        d.Cc = q->Cc;
        d.T = q->T;
        d.transposed = q->o;
        d.nplanes = q->planes;
        d.gx = (q->Cc + 31) / 32;
        d.gy = (q->R + 31) / 32;
        b.start[b.count + 1] = b.start[b.count] + (unsigned)(d.gx * d.gy * q->T);
        ++b.count;
        if (!q->ready) EG_HIP(hipEventCreateWithFlags(&q->ready, hipEventDisableTiming));
        q->built_on = c.stream;
        q->waited = 0;
        q->valid = true;
        built.push_back(q);
        if (b.count == kSplitMax) flush();
      }
      flush();
      for (LimbCache* q : built) EG_HIP(hipEventRecord(q->ready, c.stream));
    } else if (lc->built_on != c.stream && !(lc->waited & stream_bit(c))) {
      // built on another stream: this stream's reads go behind the build (later reads on it follow in stream order)
      EG_HIP(hipStreamWaitEvent(c.stream, lc->ready, 0));
      lc->waited |= stream_bit(c);
    }
    return lc->buf;
  }
  unsigned short*& tmp = transposed ? w->wlT : w->wl;
  if (!tmp) {
    tmp = reinterpret_cast<unsigned short*>(c.arena.allocf(((size_t)nplanes * n + 1) / 2));
    launch_split_bf16x3(w->w, tmp, w->T, w->R, w->Cc, o, n, nplanes, c.stream);
  }
  return tmp;
}

// the filter rounded to bf16 with the contracted index contiguous, for the 64 x 64 bf16 kernel (k_gemm_bf16.hip): per sub-step
static void bf16_filter_copy(Ctx& c, ConvArgs& a, WRef* w, bool transposed) {
  unsigned short*& slot = transposed ? w->wbT : w->wb;
  if (!slot) {
    const size_t n = (size_t)w->T * w->R * w->Cc;
    slot = reinterpret_cast<unsigned short*>(c.arena.allocf((n + 1) / 2));
    if (transposed) launch_tap_transpose_bf16(w->w, slot, w->T, w->R, w->Cc, c.stream);
    else launch_cvt_bf16(w->w, slot, n, c.stream);
    if (dbg_env("EG_TRACE_CVT")) fprintf(stderr, "[cvt] weight n=%zu transposed=%d lane=%d\n", n, (int)transposed, c.cur_lane);
  }
  a.wgt_bf = slot;
  a.wbf_ts = (long long)a.C * a.N;
  a.ldwbf = a.C;
}

static void attach_bf16(Ctx& c, ConvArgs& a, WRef* w, bool transposed) {
  c.alt_w = nullptr;   // (a hand-over to run_conv that an earlier caller never consumed must not reach this launch)
  if (c.cfg.x3) {   // three limb planes instead of one rounded copy (k_gemm_x3.hip)
    if (a.C % 32 != 0 || a.N <= 16 || a.N % 4 != 0) return;
    const size_t n = (size_t)w->T * w->R * w->Cc;
    a.wgt_bf = limb_planes(c, w, transposed, 3);
    a.wbf_ts = (long long)a.C * a.N;
    a.ldwbf = a.C;
    a.limbs = 3;
    a.wbf_plane = (long long)n;
    a.wbf_bytes = 6 * n < (1ull << 31) ? (unsigned)(6 * n) : 0u;
    return;
  }
  if (!c.cfg.bf16 || a.C % 32 != 0 || a.N <= 16 || a.N % 4 != 0) return;
  // >= 128 output channels: the ping-pong kernel in its one-plane form (same arithmetic: both operands rounded to
  // bf16, fp32 accumulation), filter as ONE K-step-major plane; narrower layers keep the 64x64 bf16 kernel
  static const bool bf16_pp = !(dbg_env("EG_BF16_PP") && atoi(dbg_env("EG_BF16_PP")) == 0);
  const size_t xbytes = (size_t)a.NB * a.IH * a.IW * a.x_ps * sizeof(float);
  if (bf16_pp && a.N >= 128 && xbytes < (1ull << 31) && a.x_ps % 4 == 0 && a.y_ps % 4 == 0) {
    const size_t n = (size_t)w->T * w->R * w->Cc;
    a.wgt_bf = limb_planes(c, w, transposed, 1);
    a.wbf_ts = (long long)a.C * a.N;
    a.ldwbf = a.C;
    a.limbs = 1;
    a.wbf_plane = (long long)n;
    a.wbf_bytes = 2 * n < (1ull << 31) ? (unsigned)(2 * n) : 0u;
    if (a.wbf_bytes) return;
    a.limbs = 0;
  }
  // 64 .. 127 output channels: the plane-fed kernel (k_conv_bf16p.hip) has a 64-channel form; whether it takes the launch is
  // only known in run_conv (K split, tile geometry) -- remember the filter so that run_conv can ask for its K-step-major plane
  if (a.N >= 64) {
    // (the rounded copy below is made in run_conv, and only if the plane-fed kernel does not take the launch: it used to be
    //  made here and then left unread -- one cvt_bf16 launch per use of g_dconv_3's filter)
    c.alt_w = w;
    c.alt_transposed = transposed;
    a.wgt_bf = reinterpret_cast<const unsigned short*>(uintptr_t(16));   // (placeholder until run_conv: the K-split model looks at non-null only)
    a.wbf_ts = (long long)a.C * a.N;
    a.ldwbf = a.C;
    return;
  }
  bf16_filter_copy(c, a, w, transposed);
}

// bf16 plane of a tensor's data / gradient (bf16 mode; common.h: Tensor::d_bf).  A plane its producer wrote is visible
// wherever the tensor is.  One made on demand lives on the stream that made it: the same stream may reuse it, and so may
// the lane's LEAF stream for a plane made on the lane's main stream (a LeafScope orders the leaf stream behind everything
// the main stream held when the scope was entered, and nothing is enqueued on the main stream inside a scope); anybody
// else -- another lane reading a shared tensor -- converts for itself.
static const unsigned short* tensor_plane(Ctx& c, T t, bool grad) {
  unsigned short*& p = grad ? t->g_bf : t->d_bf;
  hipStream_t& ps = grad ? t->g_bf_stream : t->d_bf_stream;
  const Ctx::Lane& l = c.lanes[c.cur_lane];
  if (p && (ps == nullptr || ps == c.stream || (c.stream == l.leaf && ps == l.stream))) return p;
  unsigned short* q = reinterpret_cast<unsigned short*>(c.arena.allocf((t->numel() + 1) / 2));
  launch_cvt_bf16(grad ? t->g : t->d, q, t->numel(), c.stream);
  if (dbg_env("EG_TRACE_CVT"))
    fprintf(stderr, "[cvt] tensor %s n=%d h=%d w=%d c=%d lane=%d leaf=%d had_plane=%d\n", grad ? "grad" : "data", t->n, t->h, t->w, t->c,
            c.cur_lane, (int)(c.stream == l.leaf), (int)(p != nullptr));
  if (!p) {
    p = q;
    ps = c.stream;
  }
  return q;
}

// A tensor stored as its bf16 plane only (Tensor::d_plane_only) and a reader that wants fp32 after all: widen the plane into d
// (exact; every reader of such a tensor is a dense contraction that would round to bf16 itself)
static void need_f32(Ctx& c, T t) {
  if (!t->d_plane_only) return;
  Ctx::Lane& l = c.lanes[c.cur_lane];
  if (c.stream == l.leaf && l.leaf != l.stream) {
    // inside a LeafScope (a filter gradient that gathers fp32): the widening runs on the lane's MAIN stream and the leaf stream
    // waits for it -- clearing d_plane_only below tells every later main-stream reader that d is valid, which is only true on
    // the stream that wrote it
    launch_cvt_from_bf16(t->d_bf, t->d, t->numel(), l.stream);
    EG_HIP(hipEventRecord(l.leaf_in, l.stream));
    EG_HIP(hipStreamWaitEvent(l.leaf, l.leaf_in, 0));
  } else {
    launch_cvt_from_bf16(t->d_bf, t->d, t->numel(), c.stream);
  }
  if (dbg_env("EG_TRACE_CVT")) fprintf(stderr, "[widen] data n=%d h=%d w=%d c=%d lane=%d label=%s\n", t->n, t->h, t->w, t->c, c.cur_lane, c.cur_label.c_str());
  t->d_plane_only = false;
}
void tensor_need_f32(Ctx& c, T t) { need_f32(c, t); }

// Producer-side planes: in bf16 mode an element-wise kernel whose output (or output gradient) is shaped like a dense
// contraction's operand (channels a multiple of 32) stores the bf16 plane next to the fp32 tensor -- 2 extra bytes per
// element once instead of a conversion pass (4 B read + 2 B written) per consumer.  Returns the plane to write, or null.
static unsigned short* producer_plane(Ctx& c, T t, bool grad) {
  static const bool on = dbg_env("EG_NO_PRODUCER_PLANES") == nullptr;
  if (!on || !c.cfg.bf16 || t->c % 32 != 0 || t->numel() * sizeof(unsigned short) >= (1ull << 31)) return nullptr;
  unsigned short* q = reinterpret_cast<unsigned short*>(c.arena.allocf((t->numel() + 1) / 2));
  (grad ? t->g_bf : t->d_bf) = q;
  (grad ? t->g_bf_stream : t->d_bf_stream) = nullptr;   // visible wherever the tensor's data / gradient is
  return q;
}

// launch one implicit-GEMM conv; layers too small to fill the chip with output tiles split K over blocks.
// xsrc / xgrad: the tensor whose data (gradient) a.x is -- the bf16 mode's plane-fed kernel reads its bf16 plane instead
static void run_conv(Ctx& c, ConvArgs& a, T xsrc, bool xgrad, float* x_full) {
  a.y_total = (size_t)a.NB * a.OHf * a.OWf * a.y_ps;
  a.pk = (c.cfg.x3 || c.cfg.bf16) ? 1 : 0;   // thin layers: the packed-K limb kernel (fp32 results; the exact-fp32 mode keeps conv_smallc)
  {
    const size_t xb = (size_t)a.NB * a.IH * a.IW * a.x_ps * sizeof(float) / (a.x_up ? 4 : 1);
    a.x_bytes = xb < (1ull << 31) ? (unsigned)xb : 0u;
  }
  a.ksplit = conv_pick_ksplit(a);
  // Stride-1 layers with >= 128 output channels that the cost model sent to the 64x64 limb kernel (too few pixels for the
  // ping-pong kernel's 256-row blocks) still fit the halo-staged kernel (k_gemm_x3h.hip) when its 256-pixel tiles x K
  // splits over the channel chunks give every CU a block: 168-186 vs 111-139 TFLOP/s on the discriminators' conv3 / conv4
  // input gradients and the encoder's residual convolutions.
  static const bool halo_small = dbg_env("EG_NO_HALO_SMALL") == nullptr;
  if (halo_small && !a.tuned && a.limbs == 3 && a.x3_tile == 2 && a.N >= 64 && a.S == 1) {
    const ConvArgs keep = a;
    const int bn = a.N >= 128 ? 128 : 64;
    const long long blocks = (long long)((a.NB * a.ph[0].OHp * a.ph[0].OWp + 255) / 256) * ((a.N + bn - 1) / bn) * a.nphase;
    int ks = (int)std::min<long long>(8, std::max<long long>(1, (256 + blocks - 1) / blocks));
    ks = std::min(ks, a.C / 32);
    a.x3_tile = 1;
    a.ksplit = ks;
    if (blocks * ks < 128 || !conv_x3h_eligible(a)) a = keep;
  }
#ifdef EG_EXPERIMENTS
  if (a.limbs == 3 && a.x3_tile == 1) {   // EG_X3Q=1: the launches of the one-block-per-CU halo kernel that the two-block form takes
    ConvArgs b = a;
    b.ksplit = conv_x3q_pick_ksplit(a);
    if (conv_x3q_eligible(b)) {
      a.ksplit = b.ksplit;
      a.x3q = 1;
    }
  }
#endif
  WRef* alt_w = c.alt_w;
  c.alt_w = nullptr;
  if (c.cfg.bf16 && a.limbs == 0 && alt_w && xsrc && a.x == (xgrad ? xsrc->g : xsrc->d) && a.x_ps == xsrc->c) {
    // a 64 .. 127-channel layer on the rounded-copy path: would the plane-fed kernel take it with a K-step-major filter plane?
    ConvArgs b = a;
    const size_t n = (size_t)alt_w->T * alt_w->R * alt_w->Cc, pb = xsrc->numel() * sizeof(unsigned short);
    b.limbs = 1;
    b.wbf_plane = (long long)n;
    b.wbf_bytes = 2 * n < (1ull << 31) ? (unsigned)(2 * n) : 0u;
    b.wgt_bf = reinterpret_cast<const unsigned short*>(uintptr_t(16));
    b.x_bf = reinterpret_cast<const unsigned short*>(uintptr_t(16));
    b.xbf_ps = xsrc->c;
    b.xbf_bytes = pb < (1ull << 31) ? (unsigned)pb : 0u;
    b.ksplit = 1;   // (the cost model above priced the 64x64 rounded-copy kernel; the 256-pixel x 64-channel tiles run un-split)
    if (conv_bf16p_eligible(b)) {
      a.limbs = 1;
      a.wbf_plane = b.wbf_plane;
      a.wbf_bytes = b.wbf_bytes;
      a.wgt_bf = limb_planes(c, alt_w, c.alt_transposed, 1);
      a.ksplit = 1;
    }
  }
  if (c.cfg.bf16 && a.limbs == 0 && alt_w && a.wgt_bf == reinterpret_cast<const unsigned short*>(uintptr_t(16))) bf16_filter_copy(c, a, alt_w, c.alt_transposed);   // (not taken: the 64 x 64 bf16 kernel's copy)
  if (c.cfg.bf16 && a.limbs == 1 && xsrc && a.x == (xgrad ? xsrc->g : xsrc->d) && a.x_ps == xsrc->c) {
    const size_t pb = xsrc->numel() * sizeof(unsigned short);
    a.x_bf = reinterpret_cast<const unsigned short*>(uintptr_t(16));   // (placeholder: eligibility looks at alignment only)
    a.xbf_ps = xsrc->c;
    a.xbf_bytes = pb < (1ull << 31) ? (unsigned)pb : 0u;
    ConvArgs b = a;
    b.ksplit = conv_bf16p_pick_ksplit(a);   // (the split of the plane-fed kernel, not the 256-pixel-tile model's)
    if (conv_bf16p_eligible(b)) a.ksplit = b.ksplit;
    a.x_bf = conv_bf16p_eligible(a) ? tensor_plane(c, xsrc, xgrad) : nullptr;
  }
  if (xsrc && !xgrad && !a.x_bf) need_f32(c, xsrc);   // (a plane-only operand and a kernel that gathers fp32)
  if (xsrc && xgrad && !a.x_bf && xsrc->g_plane_only && a.x == xsrc->g) a.x = grad_f32_view(c, xsrc);
  if (a.epi_soft && a.ksplit > 1) {   // a K-split launch has no epilogue: its consumer's pass sums the slabs (ConvArgs::epi_soft)
    a.epi_act = 0;
    a.epi_bf = nullptr;
    a.epi_skip_f32 = 0;
    a.keep_slabs = a.keep_slabs_soft;
  }
  a.ws = a.ksplit > 1 ? c.arena.allocf((size_t)a.ksplit * a.y_total) : nullptr;
  const char* nm;
  static const bool x3_timing = dbg_env("EG_X3_TIMING") != nullptr;   // experiment: per-phase s_memtime stamps of block 0
  if (x3_timing && a.limbs == 3 && a.ksplit <= 1) {
    a.ws = c.arena.allocf(512);
    EG_HIP(hipMemsetAsync(a.ws, 0, 2048, c.stream));
  }
  if (a.x_up && !conv_up_kernel(a)) {   // the kernel that will run gathers a full-resolution operand: write it out (Tensor::g_up)
    EG_CHECK(x_full, "run_conv: x_up without a buffer to fall back to");
    launch_pool2_bwd(a.x, x_full, 0, a.NB, a.IH, a.IW, a.x_ps, c.stream, nullptr);
    a.x = x_full;
    a.x_up = 0;
    a.x_bf = nullptr;
    const size_t xb = (size_t)a.NB * a.IH * a.IW * a.x_ps * sizeof(float);
    a.x_bytes = xb < (1ull << 31) ? (unsigned)xb : 0u;
  }
  ProfScope ps(c);
  const double fl = launch_conv_gemm(a, c.stream, &nm);
  a.epi_applied = (a.epi_act && (a.caps & CONV_CAP_EPI_ACT)) ? 1 : 0;   // (reported by the launcher of the kernel that ran)
  if (x3_timing && a.limbs == 3 && a.ksplit <= 1 && a.ws) {
    long long h[128];
    EG_HIP(hipStreamSynchronize(c.stream));
    EG_HIP(hipMemcpy(h, a.ws, sizeof h, hipMemcpyDeviceToHost));
    for (int g = 0; g < 2; ++g)
      for (int k = 0; k < 4; ++k) {
        const long long* t = h + g * 64 + k * 8;
        fprintf(stderr, "X3T grp%d k%d t0=%lld | valu_phase=%lld bar=%lld mfma=%lld bar=%lld\n", g, k, t[0] - h[0], t[1] - t[0],
                t[2] - t[1], t[3] - t[2], t[4] - t[3]);
      }
  }
  if (c.prof_on) {
    int maxM = 0, taps = 0;
    for (int p = 0; p < a.nphase; ++p) {
      maxM = std::max(maxM, a.NB * a.ph[p].OHp * a.ph[p].OWp);
      taps += a.ph[p].Ty * a.ph[p].Tx;
    }
    char buf[96];
    snprintf(buf, sizeof buf, "%d,%d,%d,%d,%d,%d", maxM, a.N, a.C, taps, a.nphase, a.ksplit);
    ps.done(nm, fl, buf);
  }
}

static void same_pad(int n, int k, int s, int* out, int* before) {
  *out = (n + s - 1) / s;
  int total = (*out - 1) * s + k - n;
  if (total < 0) total = 0;
  *before = total / 2;
}

// ---------------------------------------------------------------------------------------------
// conv2d -- edgegan/nn/modules/conv.py:13-36 (conv2d: SAME / REFLECT+VALID, optional bias) and
// conv.py:278-285 (conv2d2: SAME stride 1 + biases).  Weights HWIO = [T][Cin][Cout].
// ---------------------------------------------------------------------------------------------
static void conv_dgrad(Ctx& c, T x, T y, WRef* w, int k, int stride, int pad, int pt, int pl) {
  const int Cin = x->c, N = y->c;
  // [T][N][Cin] for the fp32 kernels; the limb / bf16 kernels take their own copies of the filter as stored
  // (a.wgt stays null then: launch_conv_gemm refuses to run an fp32 kernel without its filter)
  const bool limb_path = (c.cfg.x3 || c.cfg.bf16) && N % 32 == 0 && Cin > 16 && Cin % 4 == 0;
  const float* wT = limb_path ? nullptr : wref_transposed(c, w);
  if (c.prof_on) c.cur_label = w->name + " dgrad";
  if (y->g_up && !(stride == 1 && !(pad == PAD_REFLECT && pt > 0))) materialize_up(c, y);
  T up = y->g_up;   // y's gradient as its pooled form (Tensor::g_up): the kernel gathers from the half-resolution tensor
  if (up && !y->g) y->g = c.arena.allocf(y->numel());   // (where run_conv falls back to)
  ConvArgs a{};
  a.x = up ? up->g : y->g;
  a.x_up = up ? 1 : 0;
  a.wgt = wT;
  a.bias = nullptr;
  a.NB = x->n;
  a.IH = y->h;
  a.IW = y->w;
  a.C = N;
  a.x_ps = N;
  a.N = Cin;
  a.y_ps = Cin;
  a.KW = k;
  a.w_ts = (long long)N * Cin;
  a.ldw = Cin;
  a.reflect = 0;
  a.S = 1;
  a.D = -1;
  attach_bf16(c, a, w, false);   // rows = Cin, contracted = Cout: the filter as stored
  bool keep_ok = false;
  if (stride == 1) {
    a.kst = 1;
    a.os = 1;
    a.nphase = 1;
    if (pad == PAD_REFLECT && pt > 0) {
      // gradient w.r.t. the padded tensor, then fold the mirrored border back (tf.pad REFLECT grad)
      const int Hp = x->h + 2 * pt, Wp = x->w + 2 * pl;
      float* dpad = c.arena.allocf((size_t)x->n * Hp * Wp * Cin);
      a.y = dpad;
      a.OHf = Hp;
      a.OWf = Wp;
      a.accumulate = 0;
      a.ph[0] = ConvPhase{Hp, Wp, 0, 0, k, k, 0, 0, 0, 0};
      run_conv(c, a, y, true);
      int acc;
      float* dx = grad_acc(c, x, &acc);
      launch_reflect_fold(dpad, dx, acc, x->n, x->h, x->w, Cin, pt, c.stream);
      return;
    }
    int acc;
    a.y = grad_acc(c, x, &acc);
    a.accumulate = acc;
    a.OHf = x->h;
    a.OWf = x->w;
    a.ph[0] = ConvPhase{x->h, x->w, 0, 0, k, k, 0, 0, pt, pl};
    keep_ok = !acc;
    // x = prelu(pre) and this launch is the only writer of x's gradient: the PReLU's backward in the epilogue (un-split halo /
    // plane-fed kernel), unless the slope or a bias behind `pre` trains in this sweep (their sums come from prelu_bwd's pass)
    static const bool pepi_on = dbg_env("EG_NO_EPI_PRELU_BWD") == nullptr;
    T pre = x->prelu_pre;
    if (pepi_on && pre && !acc && !pre->g_set && !pre->g && !pre->g_inplace && !x->leaf_grad && !pre->leaf_grad &&
        !(c.param_grad_enabled && (x->prelu_leak->want_grad || pre->want_colpart)) && x->numel() < (1ull << 31)) {
      a.epi_ref = pre->d;
      a.epi_ref_bf = pre->d_plane_only ? pre->d_bf : nullptr;
      a.epi_act = ACT_PRELU_X;
      a.epi_leak = x->prelu_leak->w;
      a.epi_soft = 1;
      if (c.cfg.bf16 && x->c % 32 == 0) {
        a.epi_bf = reinterpret_cast<unsigned short*>(c.arena.allocf((x->numel() + 1) / 2));
        // d(pre) is read by the gradient kernels of the convolution behind `pre` only: the plane alone (Tensor::g_plane_only)
        static const bool gpo_on = dbg_env("EG_NO_PLANE_ONLY_GRAD") == nullptr;
        a.epi_skip_f32 = gpo_on && pre->g_plane_ok ? 1 : 0;
      }
    }
  } else {
    EG_CHECK(stride == 2 && x->h % 2 == 0 && x->w % 2 == 0, "stride-2 dgrad needs even extents");
    if (Cin == 3 && k == 4 && pt == 1 && pl == 1 && N % 16 == 0) {   // d_conv_0 / encoder stem: phase-merged form
      int acc;
      float* dx = grad_acc(c, x, &acc);
      run_merged_phases(c, y->g, x->n, y->h, y->w, N, merged_filter(c, w, 4, N), nullptr, dx, acc);
      return;
    }
    int acc;
    a.y = grad_acc(c, x, &acc);
    a.accumulate = acc;
    static const bool epi_on = dbg_env("EG_NO_EPI_ACT") == nullptr;
    if (epi_on && x->g_inplace && !acc && (x->fused_act == ACT_RELU || x->fused_act == ACT_LRELU)) {
      // x = act(conv(..)) with the activation fused in its producer (d_conv_0): this launch is the only writer of its
      // gradient, so the activation's backward rides in the epilogue (if the un-split halo kernel runs the layer)
      a.epi_ref = x->d;
      a.epi_ref_bf = x->d_plane_only ? x->d_bf : nullptr;
      a.epi_act = x->fused_act;
      a.epi_wrap = x->g_epi_wrap;
    }
    keep_ok = !acc;
    a.OHf = x->h;
    a.OWf = x->w;
    a.kst = 2;
    a.os = 2;
    a.nphase = 4;
    for (int py = 0; py < 2; ++py)
      for (int px = 0; px < 2; ++px) {
        const int kh0 = (py + pt) % 2, kw0 = (px + pl) % 2;
        ConvPhase& p = a.ph[py * 2 + px];
        p.OHp = x->h / 2;
        p.OWp = x->w / 2;
        p.ophy = py;
        p.ophx = px;
        p.Ty = (k - kh0 + 1) / 2;
        p.Tx = (k - kw0 + 1) / 2;
        p.kh0 = kh0;
        p.kw0 = kw0;
        p.offy = (py + pt - kh0) / 2;
        p.offx = (px + pl - kw0) / 2;
      }
  }
  // first writer of the gradient of a tensor whose producer's backward reads K-split slabs (Tensor::g_slab_ok): a split
  // launch leaves them un-summed
  static const bool gslab_on = dbg_env("EG_NO_GRAD_SLABS") == nullptr;
  const bool slabs_wanted = gslab_on && keep_ok && !a.accumulate && x->g_slab_ok && !x->leaf_grad && Cin % 4 == 0;
  a.keep_slabs = (slabs_wanted && !a.epi_act) ? 1 : 0;
  a.keep_slabs_soft = slabs_wanted ? 1 : 0;
  run_conv(c, a, up ? up : y, true, up ? y->g : nullptr);
  if (a.epi_act == ACT_PRELU_X && a.epi_applied && a.epi_bf) {   // the plane of what was stored = d(pre)
    x->g_bf = a.epi_bf;
    x->g_bf_stream = nullptr;
    x->g_plane_only = a.epi_skip_f32 != 0;
  }
  if (a.keep_slabs && a.ksplit > 1) {
    x->g_slabs = a.ws;
    x->g_nslab = a.ksplit;
  }
  if (!a.accumulate) x->g_act_done = a.epi_applied != 0;   // (first writer: decides anew in every backward sweep)
}

// xt / xgrad, dyt / dygrad: the tensors whose data (or gradient) the gathered and the dense operand are -- the bf16 mode's
// plane-fed kernel reads their bf16 planes
static void conv_wgrad(Ctx& c, T xt, bool xgrad, int NB, int IH, int IW, int C, T dyt, bool dygrad, int OH, int OW,
                       int N, WRef* w, int k, int stride, int pad_before, int reflect) {
  WgradArgs g{};
  T up = dygrad ? dyt->g_up : nullptr;   // the dense operand as its pooled form (Tensor::g_up)
  if (up && (dyt->c != N || dyt->h != OH || dyt->w != OW)) {
    materialize_up(c, dyt);
    up = nullptr;
  }
  T dsrc = up ? up : dyt;                // the tensor whose gradient / plane the kernel reads
  g.dy_up = up ? 1 : 0;
  const float* xg = xgrad ? xt->g : xt->d;
  const float* dy = dygrad ? dsrc->g : dyt->d;
  g.x = xg;
  g.dy = dy;
  g.dw = w->g;
  g.NB = NB;
  g.IH = IH;
  g.IW = IW;
  g.C = C;
  g.x_ps = C;
  g.OH = OH;
  g.OW = OW;
  g.N = N;
  g.dy_ps = N;
  g.KH = k;
  g.KW = k;
  g.S = stride;
  g.pad = pad_before;
  g.w_ts = (long long)C * N;
  g.ldw = N;
  g.reflect = reflect;
  g.accumulate = *w->g_set ? 1 : 0;
  g.bf16 = c.cfg.bf16 ? 1 : 0;
  if (c.cfg.x3 || c.cfg.bf16) {
    g.limbs = c.cfg.x3 ? 3 : 1;
    const size_t xb = (size_t)NB * IH * IW * C * sizeof(float), yb = (size_t)NB * OH * OW * N * sizeof(float) / (up ? 4 : 1);
    g.x_bytes = xb < (1ull << 31) ? (unsigned)xb : 0u;
    g.dy_bytes = yb < (1ull << 31) ? (unsigned)yb : 0u;
  }
  if (c.cfg.bf16 && g.limbs == 1 && xt->c == C && dyt->c == N) {
    const size_t xb = xt->numel() * sizeof(unsigned short), yb = dsrc->numel() * sizeof(unsigned short);
    g.x_bf = g.dy_bf = reinterpret_cast<const unsigned short*>(uintptr_t(16));   // (placeholders: the shape test looks at alignment only)
    g.xbf_bytes = xb < (1ull << 31) ? (unsigned)xb : 0u;
    g.dybf_bytes = yb < (1ull << 31) ? (unsigned)yb : 0u;
    if (wgrad_bf16p_units(g)) {
      g.x_bf = tensor_plane(c, xt, xgrad);
      g.dy_bf = tensor_plane(c, dsrc, dygrad);
    } else {
      g.x_bf = g.dy_bf = nullptr;
    }
  }
  if (!g.x_bf) {
    if (!xgrad) need_f32(c, xt);
    if (!dygrad) need_f32(c, dyt);
    if (dygrad && dsrc->g_plane_only) g.dy = grad_f32_view(c, dsrc);
    if (xgrad && xt->g_plane_only) g.x = grad_f32_view(c, xt);
  }
  g.nsplit = wgrad_pick_splits(g);
  const size_t wsf = wgrad_workspace_floats(g);
  g.ws = wsf ? c.arena.allocf(wsf) : nullptr;
  if (g.dy_up && !wgrad_up_kernel(g)) {   // the kernel that will run reads a full-resolution operand: write it out, for this launch
    float* full = c.arena.allocf(dyt->numel());   // (on the stream of this launch -- the leaf stream; conv_dgrad falls back on its own)
    launch_pool2_bwd(up->g, full, 0, dyt->n, dyt->h, dyt->w, dyt->c, c.stream, nullptr);
    g.dy = full;
    g.dy_up = 0;
    const size_t yb = dyt->numel() * sizeof(float);
    g.dy_bytes = yb < (1ull << 31) ? (unsigned)yb : 0u;
  }
  g.defer_reduce = c.defer_on ? 1 : 0;
  for (const RedDesc& q : c.deferred)   // a queued reduction into this filter's gradient comes first (summation order)
    if (q.dst == g.dw) {
      run_deferred(c, c.stream);
      break;
    }
  const char* nm;
  if (c.prof_on) c.cur_label = w->name + " wgrad";
  ProfScope ps(c);
  const double fl = launch_wgrad(g, c.stream, &nm);
  if (g.reduce_deferred)
    defer_reduction(c, RedDesc{g.ws, g.dw, 0, k * k, C, N, g.w_ts, g.ldw, g.nsplit, g.accumulate});
  if (c.prof_on) {
    char buf[96];
    snprintf(buf, sizeof buf, "%d,%d,%d,%d,1,%d", NB * OH * OW, N, C, k * k, g.nsplit);
    ps.done(nm, fl, buf);
  }
  *w->g_set = true;
}

static void bias_grad(Ctx& c, Param* b, const float* dy, size_t rows, int C) {
  if (c.defer_on) {   // row-block partials now (into the arena: they outlive this call), their sum with the next batch
    float* part = c.arena.allocf(col_sum_partial_floats(rows, C));
    const int nrb = launch_col_sum_stage1(dy, rows, C, part, c.stream);
    defer_reduction(c, RedDesc{part, b->g, 1, 0, C, 0, 0, 0, nrb, b->g_set ? 1 : 0});
    b->g_set = true;
    return;
  }
  if (c.scratch_floats < col_sum_scratch_floats(rows, C)) fail(1, "scratch too small for bias grad");
  launch_col_sum(dy, b->g, b->g_set ? 1 : 0, rows, C, c.scratch, c.stream);
  b->g_set = true;
}

static T prelu_tail(Ctx& c, T x, T out, Param* leak, float leak_const);

static bool need_params(WRef* w, Param* bias, Param* leak) {
  return w->want_grad || (bias && bias->want_grad) || (leak && leak->want_grad);
}

T conv2d(Ctx& c, T x, WRef* w, Param* bias, int k, int stride, int pad, T accumulate_into, int fused_act, Param* post_prelu) {
  const int Cin = x->c, N = w->Cc;
  EG_CHECK(w->T == k * k && w->R == Cin, "conv2d weight shape: T=%d R=%d vs k=%d Cin=%d", w->T, w->R, k, Cin);
  int OH, OW, pt, pl;
  if (pad == PAD_REFLECT) {
    EG_CHECK(stride == 1, "REFLECT conv is stride 1");
    pt = pl = (k - 1) / 2;
    OH = x->h;
    OW = x->w;
  } else {
    same_pad(x->h, k, stride, &OH, &pt);
    same_pad(x->w, k, stride, &OW, &pl);
  }
  T into = c.conv_out_into;   // (conv_into: forward only, into a caller's tensor)
  c.conv_out_into = nullptr;
  T y = accumulate_into ? accumulate_into : (into ? into : new_tensor(c, x->n, OH, OW, N, false));
  EG_CHECK(y->n == x->n && y->h == OH && y->w == OW && y->c == N, "conv2d output shape");
  if (accumulate_into) y->d_bf = nullptr;   // (a bf16 plane of the value this launch adds to is stale afterwards)
  ConvArgs a{};
  a.x = x->d;
  a.x_is_activation = 1;
  a.wgt = w->w;
  a.bias = bias ? bias->w : nullptr;
  a.y = y->d;
  a.NB = x->n;
  a.IH = x->h;
  a.IW = x->w;
  a.C = Cin;
  a.x_ps = Cin;
  a.OHf = OH;
  a.OWf = OW;
  a.N = N;
  a.y_ps = N;
  a.KW = k;
  a.kst = 1;
  a.S = stride;
  a.D = 1;
  a.os = 1;
  a.w_ts = (long long)Cin * N;
  a.ldw = N;
  a.reflect = pad == PAD_REFLECT ? 1 : 0;
  a.accumulate = accumulate_into ? 1 : 0;
  a.nphase = 1;
  a.ph[0] = ConvPhase{OH, OW, 0, 0, k, k, 0, 0, -pt, -pl};
  attach_bf16(c, a, w, true);    // rows = Cout, contracted = Cin
  a.pk = (c.cfg.x3 || c.cfg.bf16) ? 1 : 0;   // thin layers: the packed-K limb kernel (fp32 results; the exact-fp32 mode keeps conv_smallc)
  // conv + ReLU / LeakyReLU in one pass where the small-C kernel runs the layer (its epilogue applies the activation; the
  // backward turns dL/d(out) into dL/d(pre) in place from the OUTPUT's sign, like act()): d_conv_0, the encoder stem
  bool fused = false;
  if (fused_act == ACT_RELU || fused_act == ACT_LRELU) {
    static const bool fuse_on = dbg_env("EG_NO_FUSED_ACT") == nullptr;
    a.y_total = (size_t)a.NB * a.OHf * a.OWf * a.y_ps;
    fused = fuse_on && !accumulate_into && (conv_pk_eligible(a) || (!conv_scatter_eligible(a) && conv_smallc_eligible(a)));
    if (fused) a.act = fused_act;
    // bf16 mode: the packed-K kernel also stores the activated output's plane (N % 4 == 0 and a 4-aligned pitch: its float4 path)
    static const bool pk_plane = dbg_env("EG_NO_PK_PLANE") == nullptr;
    if (fused && pk_plane && conv_pk_eligible(a) && N % 4 == 0 && a.y_ps % 4 == 0)
      a.y_bf = (into && y->d_bf) ? y->d_bf : producer_plane(c, y, false);   // (a view's rows of its parent's plane)
    // Ctx::conv_plane_only_next: every reader of this output takes the plane (the next convolution, its filter gradient) or only the
    // SIGN of the activated value (the fused activation's backward: ConvArgs::epi_ref_bf, launch_act_bwd_bf) -- no fp32 copy
    static const bool po0_on = dbg_env("EG_NO_PLANE_ONLY_CONV") == nullptr;
    if (a.y_bf && c.conv_plane_only_next && po0_on) {
      a.y_skip = 1;
      y->d_plane_only = true;
    }
  }
  c.conv_plane_only_next = false;
  if (c.prof_on) c.cur_label = w->name + " fwd";
  const bool need = !into && (x->needs_grad || w->want_grad || (bias && bias->want_grad));
  // convolution read by an InstanceNorm alone (Ctx::feeds_in): a K-split launch keeps its slabs, the norm's load pass sums
  // them (bias + slabs in conv_split_reduce's order: same bits, one launch and one round trip less) and, in training,
  // stores the sum as this tensor's value for the backward passes (in_act)
  static const bool slab_fusion = dbg_env("EG_NO_SLAB_IN") == nullptr;
  static const bool slab_train = dbg_env("EG_NO_SLAB_IN_TRAIN") == nullptr;   // A/B: the fusion in training graphs
  static const bool slab_flat = dbg_env("EG_NO_SLAB_FLAT") == nullptr;        // A/B: the hand-over to prelu / pool2_add / minmax
  const int feeds = post_prelu ? (int)FEED_FLAT : c.feeds_in;
  c.feeds_in = FEED_NONE;
  bool slabs_ok = false;
  if (feeds == FEED_IN)
    slabs_ok = slab_fusion && (!need || slab_train) && !accumulate_into && !fused_act && N % 32 == 0 && OH * OW <= 1024;
  else if (feeds == FEED_FLAT)   // prelu / pool2_add read any [..][N] tensor quad by quad
    slabs_ok = slab_flat && !accumulate_into && !fused_act && N % 4 == 0 && y->numel() < (1ull << 31);
  else if (feeds == FEED_MINMAX)   // the gate's hidden-state convolution adds to the image-row convolution's output (the base)
    slabs_ok = slab_flat && !fused_act && N % 8 == 0;
  a.keep_slabs = slabs_ok ? 1 : 0;
  // conv -> PReLU (conv.py:217-220): second output of the epilogue where the kernel has one
  T y2 = nullptr;
  unsigned short* y2_plane = nullptr;
  static const bool epi2_on = dbg_env("EG_NO_EPI_PRELU") == nullptr;
  if (post_prelu) {
    EG_CHECK(!accumulate_into && !fused_act, "conv2d: post_prelu excludes accumulate_into / fused_act");
    y2 = new_tensor(c, x->n, OH, OW, N, false);
    y2_plane = producer_plane(c, y2, false);
    static const bool plane_only_on = dbg_env("EG_NO_PLANE_ONLY") == nullptr;
    y2->d_plane_only = plane_only_on && c.prelu_plane_only && y2_plane != nullptr;
    if (epi2_on && N % 4 == 0) {
      a.y2_leak = post_prelu->w;
      a.y2 = y2->d_plane_only ? nullptr : y2->d;
      a.y2_bf = y2_plane;
      // bf16 mode, nothing behind this convolution trains in this sweep (the classifier inside the generator sub-steps): the
      // pre-activation is read for the SIGN only (PReLU backward: ConvArgs::epi_ref_bf, or prelu_bwd after need_f32) -- its plane alone
      static const bool po1_on = dbg_env("EG_NO_PLANE_ONLY_CONV") == nullptr;
      if (po1_on && c.cfg.bf16 && y2->d_plane_only && !need_params(w, bias, post_prelu) && N % 32 == 0 && y->numel() * 2 < (1ull << 31)) {
        a.epi_bf = reinterpret_cast<unsigned short*>(c.arena.allocf((y->numel() + 1) / 2));
        a.epi_skip_f32 = 1;
      }
    }
  }
  c.prelu_plane_only = false;
  Ctx::PoolEpi& pe = c.pool_epi;
  static const bool pool_epi_on = dbg_env("EG_NO_POOL_EPI") == nullptr;
  if (pe.out && pool_epi_on && !accumulate_into && !fused_act && !post_prelu && stride == 1) {
    a.pool_y = pe.out->d;
    a.pool_b = pe.b->d;
    a.pool_leak = pe.out2 ? pe.leak->w : nullptr;
    a.pool_y2 = pe.out2 && !pe.out2->d_plane_only ? pe.out2->d : nullptr;
    a.pool_y2_bf = pe.out2 ? pe.plane2 : nullptr;
  }
  run_conv(c, a, x, false);
  if (a.epi_skip_f32 && a.epi_bf && (a.caps & CONV_CAP_Y2)) {   // (the kernels with the second store are the ones with this one)
    y->d_bf = a.epi_bf;
    y->d_bf_stream = nullptr;
    y->d_plane_only = true;
  }
  pe.done = a.pool_y && (a.caps & CONV_CAP_POOL);
  if (slabs_ok && a.ksplit > 1) {
    y->slabs = a.ws;
    y->nslab = a.ksplit;
    y->slab_bias = a.bias;
    y->slab_base = accumulate_into ? y->d : nullptr;
  }
  if (post_prelu) {
    if (y->nslab > 1) {   // the PReLU pass sums the slabs (and stores the sum: its backward reads the pre-activation)
      const SlabSrc xs = data_src(y);
      launch_prelu_fwd(y->d, post_prelu->w, 0.f, y2->d_plane_only ? nullptr : y2->d, y->numel(), c.stream, y2_plane, &xs, y->d);
      y->nslab = 0;
    } else if (!(a.y2_leak && (a.caps & CONV_CAP_Y2))) {
      launch_prelu_fwd(y->d, post_prelu->w, 0.f, y2->d_plane_only ? nullptr : y2->d, y->numel(), c.stream, y2_plane);
    }
  }
  if (need) {
    y->needs_grad = true;
    y->g_inplace = fused;
    y->fused_act = fused ? fused_act : 0;
    // (the closure below reads y's gradient through planes -- where its kernels are the plane-fed ones: not the thin layers'
    //  (3 / 8 input channels), not the REFLECT-padded input gradient, which goes through the padded tensor)
    y->g_plane_ok = c.cfg.bf16 && k > 1 && N % 64 == 0 && Cin % 32 == 0 && pad != PAD_REFLECT && !fused && !accumulate_into;
    if (bias && bias->want_grad && !accumulate_into) y->want_colpart = true;   // (the writer of y's gradient may sum its columns)
    Ctx* cp = &c;
    c.tape.push([=]() {
      Ctx& c = *cp;
      if (!y->g_set) return;
      if (fused && !y->g_act_done) {
        materialize_up(c, y);
        need_f32(c, y);
        launch_act_bwd(y->d, y->g, y->g, 0, y->numel(), fused_act, c.stream);   // in place: g *= act'(out)
        y->g_bf = nullptr;           // (a plane its writer stored holds the un-masked values)
        y->g_colsum_src = nullptr;
        y->g_colpart = nullptr;
      }
      // bf16 mode: the plane of y's gradient serves the filter gradient (leaf stream) AND the input gradient (this stream):
      // if y's producer left none, make it here, before the leaf scope opens -- both streams may then read it.  Only the
      // multi-tap layers have plane-fed kernels (k_conv_bf16p.hip: 2 .. 3 taps per dimension; k_wgrad_bf16p.hip: 3x3, 4x4,
      // 5x5): a 1x1 layer would pay the conversion for nothing
      if (c.cfg.bf16 && k > 1 && y->c % 64 == 0 && w->want_grad && x->needs_grad && c.param_grad_enabled)
        (void)tensor_plane(c, y->g_up ? y->g_up : y, true);   // (Tensor::g_up: the pooled gradient is what both kernels read)
      if (c.param_grad_enabled && (w->want_grad || (bias && bias->want_grad))) {
        LeafScope leaf(c);
        if (w->want_grad)
          conv_wgrad(c, x, false, x->n, x->h, x->w, Cin, y, true, OH, OW, N, w, k, stride, pt, pad == PAD_REFLECT);
        if (bias && bias->want_grad) bias_grad_of(c, bias, y);
      }
      if (x->needs_grad) {
        if (x->leaf_grad) {   // gradient towards a leaf-side tensor (the classifier's image pyramid): off the dependent chain
          LeafScope leaf(c);
          conv_dgrad(c, x, y, w, k, stride, pad, pt, pl);
        } else {
          conv_dgrad(c, x, y, w, k, stride, pad, pt, pl);
        }
      }
    });
  }
  if (into) {
    y->fused_act = fused ? fused_act : 0;
    c.conv_wrote_plane = a.y_bf != nullptr;
    return y;
  }
  if (fused_act && !fused) return act(c, y, fused_act);
  if (post_prelu) return prelu_tail(c, y, y2, post_prelu, 0.f);
  return y;
}

T view_rows(Ctx& c, T parent, int n0, int n) {
  EG_CHECK(n0 >= 0 && n > 0 && n0 + n <= parent->n, "view_rows [%d, %d) of %d images", n0, n0 + n, parent->n);
  const size_t off = (size_t)n0 * parent->h * parent->w * parent->c;
  T v = new_tensor(c, n, parent->h, parent->w, parent->c, false, false);
  v->d = parent->d + off;
  if (!parent->g) parent->g = c.arena.allocf(parent->numel());
  v->g = parent->g + off;
  if (parent->d_bf) {
    v->d_bf = parent->d_bf + off;
    v->d_bf_stream = parent->d_bf_stream;
    v->d_plane_only = parent->d_plane_only;   // (a reader of the view that wants fp32 widens the view's rows: need_f32)
  }
  if (parent->g_bf) {
    v->g_bf = parent->g_bf + off;
    v->g_bf_stream = parent->g_bf_stream;
    v->g_plane_only = parent->g_plane_only;
  }
  return v;
}

int conv_into(Ctx& c, T x, WRef* w, T y, int k, int stride, int fused_act) {
  c.conv_out_into = y;
  c.feeds_in = FEED_NONE;
  c.conv_wrote_plane = false;
  conv2d(c, x, w, nullptr, k, stride, PAD_SAME, nullptr, fused_act);
  const int r = (y->fused_act != 0 ? 1 : 0) | (c.conv_wrote_plane ? 2 : 0);
  y->fused_act = 0;   // (the caller applies / differentiates the activation itself)
  return r;
}

void conv_backward_w(Ctx& c, T x, T y, WRef* w, int k, int stride) {
  int OH, OW, pt, pl;
  same_pad(x->h, k, stride, &OH, &pt);
  same_pad(x->w, k, stride, &OW, &pl);
  // (a plane made on the main stream is read by both gradients; a thin filter gradient -- 3 / 6 input channels -- reads fp32)
  if (c.cfg.bf16 && k > 1 && y->c % 64 == 0 && x->c % 32 == 0) (void)tensor_plane(c, y, true);
  LeafScope leaf(c);
  conv_wgrad(c, x, false, x->n, x->h, x->w, x->c, y, true, OH, OW, y->c, w, k, stride, pt, 0);
}

void linear1_backward_w(Ctx& c, const float* x, const float* dy, Param* w, Param* bias, int bias_rows, int rows, int K) {
  LeafScope leaf(c);
  launch_gemv_bwd_w(x, dy, w->g, w->g_set ? 1 : 0, bias->g, bias->g_set ? 1 : 0, bias_rows, K, c.stream);
  if (rows > bias_rows)
    launch_gemv_bwd_w(x + (size_t)bias_rows * K, dy + bias_rows, w->g, 1, nullptr, 0, rows - bias_rows, K, c.stream);
  w->g_set = bias->g_set = true;
}

void conv_backward_x(Ctx& c, T x, T y, WRef* w, int k, int stride) {
  int OH, OW, pt, pl;
  same_pad(x->h, k, stride, &OH, &pt);
  same_pad(x->w, k, stride, &OW, &pl);
  conv_dgrad(c, x, y, w, k, stride, PAD_SAME, pt, pl);
}

// ---------------------------------------------------------------------------------------------
// deconv2d -- conv.py:39-58: tf.nn.conv2d_transpose SAME, 5x5 stride 2, filter [kh,kw,Cout,Cin],
// + bias; y[o] += x[i] w[k] with o = 2i + k - 1 (Appendix B.2), run as 4 sub-pixel phases.
// ---------------------------------------------------------------------------------------------
T deconv2d(Ctx& c, T x, WRef* w, Param* bias, int cout, int fused_act, float* y_into) {
  const int k = 5, pt = 1, Cin = x->c;
  EG_CHECK(w->T == 25 && w->R == cout && w->Cc == Cin, "deconv2d weight shape");
  const int OH = x->h * 2, OW = x->w * 2;
  T y = y_into ? wrap_tensor(c, y_into, x->n, OH, OW, cout) : new_tensor(c, x->n, OH, OW, cout, false);
  const bool merged = cout == 3 && Cin % 16 == 0;   // g_dconv_4: all 4 phases in one 3x3 launch over the coarse grid
  if (merged) {
    if (c.prof_on) c.cur_label = w->name + " fwd";
    // forward-only graphs (inference, the generator outputs of the discriminator sub-steps): the output tanh in the store
    const bool fwd_only = !(x->needs_grad || w->want_grad || (bias && bias->want_grad));
    static const bool tanh_on = dbg_env("EG_NO_EPI_TANH") == nullptr;
    const bool done = run_merged_phases(c, x->d, x->n, x->h, x->w, Cin, merged_filter(c, w, 5, Cin), bias ? bias->w : nullptr, y->d, 0,
                                        (fwd_only && tanh_on && fused_act == ACT_TANH) ? ACT_TANH : 0);
    y->fused_act = done ? ACT_TANH : 0;
  }
  const float* wT = merged ? nullptr : wref_transposed(c, w);  // [25][Cin][Cout]
  ConvArgs a{};
  a.x = x->d;
  a.x_is_activation = 1;
  a.wgt = wT;
  a.bias = bias ? bias->w : nullptr;
  a.y = y->d;
  a.NB = x->n;
  a.IH = x->h;
  a.IW = x->w;
  a.C = Cin;
  a.x_ps = Cin;
  a.OHf = OH;
  a.OWf = OW;
  a.N = cout;
  a.y_ps = cout;
  a.KW = k;
  a.kst = 2;
  a.S = 1;
  a.D = -1;
  a.os = 2;
  a.w_ts = (long long)Cin * cout;
  a.ldw = cout;
  a.nphase = 4;
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      const int kh0 = (py + pt) % 2, kw0 = (px + pt) % 2;
      a.ph[py * 2 + px] = ConvPhase{x->h, x->w, py, px, (k - kh0 + 1) / 2, (k - kw0 + 1) / 2, kh0, kw0,
                                    (py + pt - kh0) / 2, (px + pt - kw0) / 2};
    }
  attach_bf16(c, a, w, false);   // filter [25][Cout][Cin]: rows = Cout, contracted = Cin as stored
  if (c.prof_on) c.cur_label = w->name + " fwd";
  const bool need = x->needs_grad || w->want_grad || (bias && bias->want_grad);
  static const bool slab_fusion = dbg_env("EG_NO_SLAB_IN") == nullptr;
  static const bool slab_train = dbg_env("EG_NO_SLAB_IN_TRAIN") == nullptr;
  const bool slabs_ok = c.feeds_in == FEED_IN && slab_fusion && (!need || slab_train) && !merged && cout % 32 == 0 && OH * OW <= 1024;   // see conv2d
  c.feeds_in = FEED_NONE;
  a.keep_slabs = slabs_ok ? 1 : 0;
  if (!merged) run_conv(c, a, x, false);
  if (slabs_ok && a.ksplit > 1) {
    y->slabs = a.ws;
    y->nslab = a.ksplit;
    y->slab_bias = a.bias;
  }
  if (need) {
    y->needs_grad = true;
    if (bias && bias->want_grad) y->want_colpart = true;
    Ctx* cp = &c;
    c.tape.push([=]() {
      Ctx& c = *cp;
      if (!y->g_set) return;
      if (c.param_grad_enabled && (w->want_grad || (bias && bias->want_grad))) {
        LeafScope leaf(c);
        // dW[k][co][ci] = sum_i dy[2i+k-1][co] x[i][ci]: wgrad with the roles of x and dy swapped
        if (w->want_grad) conv_wgrad(c, y, true, x->n, OH, OW, cout, x, false, x->h, x->w, Cin, w, 5, 2, 1, 0);
        if (bias && bias->want_grad) bias_grad_of(c, bias, y);
      }
      if (x->needs_grad) {
        // dx[i] = sum_k dy[2i+k-1] w[k]: a stride-2 SAME conv of dy with the filter as stored
        ConvArgs b{};
        int acc;
        b.y = grad_acc(c, x, &acc);
        b.accumulate = acc;
        b.x = y->g;
        b.wgt = w->w;
        b.NB = x->n;
        b.IH = OH;
        b.IW = OW;
        b.C = cout;
        b.x_ps = cout;
        b.OHf = x->h;
        b.OWf = x->w;
        b.N = Cin;
        b.y_ps = Cin;
        b.KW = 5;
        b.kst = 1;
        b.S = 2;
        b.D = 1;
        b.os = 1;
        b.w_ts = (long long)cout * Cin;
        b.ldw = Cin;
        b.nphase = 1;
        b.ph[0] = ConvPhase{x->h, x->w, 0, 0, 5, 5, 0, 0, -1, -1};
        attach_bf16(c, b, w, true);    // rows = Cin, contracted = Cout
        if (c.prof_on) c.cur_label = w->name + " dgrad";
        run_conv(c, b, y, true);
      }
    });
  }
  return y;
}

// linear.py:10-31 / :79-92 / :58-65: y = x W + b as a 1x1 "conv" over B rows
T linear(Ctx& c, T x, WRef* w, Param* bias) {
  EG_CHECK(x->h == 1 && x->w == 1, "linear expects [B,1,1,in]");
  return conv2d(c, x, w, bias, 1, 1, PAD_SAME);
}

// zero-copy view with shared gradient storage (tf.reshape); src must be single-use
T reshape(Ctx& c, T src, int n, int h, int w, int ch) {
  EG_CHECK((size_t)n * h * w * ch == src->numel(), "reshape numel");
  T v = new_tensor(c, n, h, w, ch, src->needs_grad, false);
  v->d = src->d;
  if (src->needs_grad) {
    if (!src->g) src->g = c.arena.allocf(src->numel());
    v->g = src->g;
    c.tape.push([=]() {
      if (v->g_set) src->g_set = true;
    });
  }
  return v;
}

// d_linear_5 (discriminator.py:74): N == 1 head on the flattened (h,w,c) activations
T linear1(Ctx& c, T x, Param* w, Param* bias) {
  const int K = (int)(x->numel() / x->n);
  EG_CHECK((size_t)K == w->numel, "linear1 K mismatch %d vs %zu", K, w->numel);
  T y = new_tensor(c, x->n, 1, 1, 1, false);
  launch_gemv_fwd(x->d, w->w, bias ? bias->w : nullptr, y->d, x->n, K, c.stream);
  const bool need = x->needs_grad || w->want_grad;
  if (need) {
    y->needs_grad = true;
    Ctx* cp = &c;
    c.tape.push([=]() {
      Ctx& c = *cp;
      if (!y->g_set) return;
      if (c.param_grad_enabled && w->want_grad) {
        const bool wb = bias && bias->want_grad;
        launch_gemv_bwd_w(x->d, y->g, w->g, w->g_set ? 1 : 0, wb ? bias->g : nullptr, wb && bias->g_set ? 1 : 0,
                          x->n, K, c.stream);
        w->g_set = true;
        if (wb) bias->g_set = true;
      }
      if (x->needs_grad) {
        int acc;
        float* dx = grad_acc(c, x, &acc);
        launch_gemv_bwd_x(y->g, w->w, dx, acc, x->n, K, c.stream);
      }
    });
  }
  return y;
}

// normalization.py:13-18 + activation.py:4-15 (fused)
// scratch of the chunked InstanceNorm kernels (maps of more than 1,024 pixels: kernels.h in_chunked_shape), from the active arena
float* in_chunk_scratch(Ctx& c, int NB, int HW, int C) {
  return in_chunked_shape(HW, C) ? c.arena.allocf(in_chunk_scratch_floats(NB, HW, C)) : nullptr;
}

T in_act(Ctx& c, T x, int a, T* stats_out, bool plane_only) {
  T out = new_tensor(c, x->n, x->h, x->w, x->c, x->needs_grad);
  T st = new_tensor(c, x->n, 1, 1, 2 * x->c, false);
  const int HW = x->h * x->w;
  // plane_only (bf16 mode, the caller knows every reader is a dense contraction): the output is stored as its bf16 plane alone --
  // the backward recomputes act'(out) from x (k_norm.hip: actg_from_x), a reader that wants fp32 widens the plane (need_f32)
  unsigned short* plane = producer_plane(c, out, false);
  static const bool po_on = dbg_env("EG_NO_PLANE_ONLY_IN") == nullptr;
  out->d_plane_only = po_on && plane_only && plane != nullptr;
  float* od = out->d_plane_only ? nullptr : out->d;
  EG_CHECK(x->nslab <= 1 || (x->c % 32 == 0 && HW <= 1024), "in_act: slab input needs C %% 32 == 0 and HW <= 1024");
  if (x->nslab > 1) {   // a K-split convolution output that was left as slabs for this pass (conv2d); in a training graph
                        // the pass also stores the sum into x->d: from here on x is an ordinary tensor
    launch_in_fwd(x->slabs, od, st->d, x->n, HW, x->c, a, c.stream, x->nslab, x->numel(), x->slab_bias,
                  x->needs_grad ? x->d : nullptr, plane);
    if (x->needs_grad) x->nslab = 0;
  } else
    launch_in_fwd(x->d, od, st->d, x->n, HW, x->c, a, c.stream, 0, 0, nullptr, nullptr, plane, in_chunk_scratch(c, x->n, HW, x->c));
  if (stats_out) *stats_out = st;
  if (x->needs_grad) {
    out->g_slab_ok = in_bwd_takes_slabs(HW, x->c);
    Ctx* cp = &c;
    c.tape.push([=]() {
      Ctx& c = *cp;
      if (!out->g_set) return;
      int acc;
      float* dx = grad_acc(c, x, &acc);
      const SlabSrc ds = grad_src(out);
      const int prow = in_bwd_colpart_rows(x->n, HW, x->c);
      float* part = colpart_for(c, x, acc, prow);
      unsigned short* gplane = producer_plane(c, x, true);
      launch_in_bwd(x->d, out->d, out->g, st->d, dx, plane_only_acc(c, x, acc, gplane), x->n, HW, x->c, a, c.stream, gplane,
                    ds.n > 1 ? &ds : nullptr, part, in_chunk_scratch(c, x->n, HW, x->c));
      colpart_done(x, part, prow);
    });
  }
  return out;
}

// forward-mode tangent of in_act at the primal (x_prim, out_prim, stats_prim)
T in_act_tangent(Ctx& c, T tx, T x_prim, T out_prim, T stats_prim, int a) {
  T tout = new_tensor(c, tx->n, tx->h, tx->w, tx->c, true);
  const int HW = tx->h * tx->w;
  launch_in_tangent_fwd(x_prim->d, out_prim->d, stats_prim->d, tx->d, tout->d, tx->n, HW, tx->c, a, c.stream,
                        producer_plane(c, tout, false), in_chunk_scratch(c, tx->n, HW, tx->c));
  Ctx* cp = &c;
  c.tape.push([=]() {
    Ctx& c = *cp;
    if (!tout->g_set) return;
    int acc_t, acc_x;
    float* dtx = grad_acc(c, tx, &acc_t);
    float* dx = grad_acc(c, x_prim, &acc_x);
    launch_in_tangent_bwd(x_prim->d, out_prim->d, stats_prim->d, tx->d, tout->g, dtx, acc_t, dx, acc_x, tx->n, HW,
                          tx->c, a, c.stream, nullptr, in_chunk_scratch(c, tx->n, HW, tx->c));
  });
  return tout;
}

T act(Ctx& c, T x, int a) {
  T out = new_tensor(c, x->n, x->h, x->w, x->c, x->needs_grad);
  launch_act_fwd(x->d, out->d, x->numel(), a, c.stream);
  if (x->needs_grad) {
    Ctx* cp = &c;
    c.tape.push([=]() {
      Ctx& c = *cp;
      if (!out->g_set) return;
      int acc;
      float* dx = grad_acc(c, x, &acc);
      launch_act_bwd(out->d, out->g, dx, acc, x->numel(), a, c.stream);
    });
  }
  return out;
}

// tangent of a piecewise-linear activation: tout = act'(out_prim) * tx
T act_tangent(Ctx& c, T tx, T out_prim, int a) {
  T tout = new_tensor(c, tx->n, tx->h, tx->w, tx->c, true);
  launch_act_bwd(out_prim->d, tx->d, tout->d, 0, tx->numel(), a, c.stream);
  Ctx* cp = &c;
  c.tape.push([=]() {
    Ctx& c = *cp;
    if (!tout->g_set) return;
    int acc;
    float* dtx = grad_acc(c, tx, &acc);
    launch_act_bwd(out_prim->d, tout->g, dtx, acc, tx->numel(), a, c.stream);
  });
  return tout;
}

// generator.py:51-52 (quirk N1): contrib batch_norm, batch statistics, then ReLU
T bn_relu(Ctx& c, T x, Param* gamma, Param* beta, Param* mm, Param* mv, int update_times, T* stats_out) {
  const int C = x->c, R = (int)x->pixels();
  T out = new_tensor(c, x->n, x->h, x->w, C, true);
  T st = new_tensor(c, 1, 1, 1, 2 * C, false);
  // (R >= 4,096 rows -- configs[4]'s generator head, 16 x 16 x 32 positions -- : the chunked two-stage BatchNorm kernels of the block
  //  norms, k_norm.hip; the one-block-per-32-channels form walks the rows in a loop: 88 us per launch there)
  const bool big = R >= 4096;
  if (big) {
    launch_bn_act_fwd(x->d, gamma->w, beta->w, out->d, st->d, c.arena.allocf(bn_scratch_floats(R, C)), nullptr, nullptr, 0, R, C, 1e-5f,
                      ACT_RELU, c.stream);
  } else {
    // (bf16 mode: the plane the first transposed convolution stages, from the same pass -- no cvt_bf16 launch behind it)
    unsigned short* plane = C % 32 == 0 ? producer_plane(c, out, false) : nullptr;
    if (!launch_bn_relu_fwd(x->d, gamma->w, beta->w, out->d, st->d, R, C, 1e-5f, c.stream, plane)) out->d_bf = nullptr;
  }
  if (stats_out) *stats_out = st;
  if (update_times > 0) launch_bn_moving_update(mm->w, mv->w, st->d, C, R, 0.9f, update_times, c.stream);
  const bool need = x->needs_grad || gamma->want_grad;
  out->needs_grad = need;
  if (need) {
    Ctx* cp = &c;
    c.tape.push([=]() {
      Ctx& c = *cp;
      if (!out->g_set) return;
      float* dx = nullptr;
      int acc = 0;
      if (x->needs_grad) dx = grad_acc(c, x, &acc);
      const bool pg = c.param_grad_enabled && gamma->want_grad;
      if (big)
        launch_bn_act_bwd(x->d, out->d, out->g, gamma->w, st->d, c.arena.allocf(bn_scratch_floats(R, C)), dx, acc, pg ? gamma->g : nullptr,
                          gamma->g_set ? 1 : 0, pg ? beta->g : nullptr, beta->g_set ? 1 : 0, R, C, 1e-5f, ACT_RELU, c.stream);
      else
        launch_bn_relu_bwd(x->d, out->d, out->g, gamma->w, st->d, dx, acc, pg ? gamma->g : nullptr,
                           gamma->g_set ? 1 : 0, pg ? beta->g : nullptr, beta->g_set ? 1 : 0, R, C, 1e-5f, c.stream);
      if (pg) gamma->g_set = beta->g_set = true;
    });
  }
  return out;
}

// normalization.py:19-25 as the norm of a conv_block / deconv_block / residual (E_norm / G_norm / D_norm = 'batch'): contrib
// batch_norm over all rows of this evaluation, learned scale / shift, then the block's activation.  fp32 in every mode (a
// plane-fed consumer converts for itself: tensor_plane).  stats_out: [mean | var] for the tangent pass.
T bn_act(Ctx& c, T x, Param* gamma, Param* beta, Param* mm, Param* mv, int a, int update_times, T* stats_out) {
  const int C = x->c, R = (int)x->pixels();
  need_f32(c, x);
  T out = new_tensor(c, x->n, x->h, x->w, C, true);
  T st = new_tensor(c, 1, 1, 1, 2 * C, false);
  launch_bn_act_fwd(x->d, gamma->w, beta->w, out->d, st->d, c.arena.allocf(bn_scratch_floats(R, C)), mm ? mm->w : nullptr,
                    mv ? mv->w : nullptr, update_times, R, C, 1e-5f, a, c.stream);
  if (stats_out) *stats_out = st;
  const bool need = x->needs_grad || gamma->want_grad;
  out->needs_grad = need;
  if (need) {
    Ctx* cp = &c;
    c.tape.push([=]() {
      Ctx& c = *cp;
      if (!out->g_set) return;
      float* dx = nullptr;
      int acc = 0;
      if (x->needs_grad) dx = grad_acc(c, x, &acc);
      const bool pg = c.param_grad_enabled && gamma->want_grad;
      launch_bn_act_bwd(x->d, out->d, out->g, gamma->w, st->d, c.arena.allocf(bn_scratch_floats(R, C)), dx, acc, pg ? gamma->g : nullptr,
                        gamma->g_set ? 1 : 0, pg ? beta->g : nullptr, beta->g_set ? 1 : 0, R, C, 1e-5f, a, c.stream);
      if (pg) gamma->g_set = beta->g_set = true;
    });
  }
  return out;
}

// forward-mode tangent of bn_act at the primal (x_prim, out_prim, stats_prim); its reverse yields the gradients w.r.t. the
// tangent, the primal input and gamma (the gradient penalty through a BatchNorm discriminator)
T bn_act_tangent(Ctx& c, T tx, T x_prim, T out_prim, T stats_prim, Param* gamma, int a) {
  const int C = tx->c, R = (int)tx->pixels();
  T tout = new_tensor(c, tx->n, tx->h, tx->w, C, true);
  launch_bn_act_tangent_fwd(x_prim->d, out_prim->d, gamma->w, stats_prim->d, tx->d, tout->d, c.arena.allocf(bn_scratch_floats(R, C)), R, C,
                            1e-5f, a, c.stream);
  Ctx* cp = &c;
  c.tape.push([=]() {
    Ctx& c = *cp;
    if (!tout->g_set) return;
    int acc_t, acc_x;
    float* dtx = grad_acc(c, tx, &acc_t);
    float* dx = grad_acc(c, x_prim, &acc_x);
    const bool pg = c.param_grad_enabled && gamma->want_grad;
    launch_bn_act_tangent_bwd(x_prim->d, out_prim->d, gamma->w, stats_prim->d, tx->d, tout->g, c.arena.allocf(bn_scratch_floats(R, C)), dtx,
                              acc_t, dx, acc_x, pg ? gamma->g : nullptr, gamma->g_set ? 1 : 0, R, C, 1e-5f, a, c.stream);
    if (pg) gamma->g_set = true;
  });
  return tout;
}

T add(Ctx& c, T a, T b, bool relu) {
  T out = new_tensor(c, a->n, a->h, a->w, a->c, a->needs_grad || b->needs_grad);
  launch_add(a->d, b->d, out->d, relu ? 1 : 0, a->numel(), c.stream);
  if (out->needs_grad) {
    Ctx* cp = &c;
    c.tape.push([=]() {
      Ctx& c = *cp;
      if (!out->g_set) return;
      // d(a) = d(b) (= d(out) masked by the ReLU): computed once; inputs whose gradient is still unset share the buffer
      float* shared = nullptr;
      for (T in : {a, b}) {
        if (!in->needs_grad) continue;
        if (!in->g_set && !in->g && shared && !in->g_inplace) {
          in->g = shared;
          in->g_set = true;
          continue;
        }
        const bool fresh = !in->g_set && !in->g;
        int acc;
        float* d = grad_acc(c, in, &acc);
        if (relu)
          launch_act_bwd(out->d, out->g, d, acc, in->numel(), ACT_RELU, c.stream);
        else
          launch_axpy(out->g, d, 1.f, acc, in->numel(), c.stream);
        if (fresh && !shared && !in->g_inplace) shared = d;
      }
    });
  }
  return out;
}

// pooling.py:4-8 / encoder.py:68
T pool2(Ctx& c, T x) {
  EG_CHECK(x->h % 2 == 0 && x->w % 2 == 0, "pool2 needs even extents");
  T out = new_tensor(c, x->n, x->h / 2, x->w / 2, x->c, x->needs_grad);
  launch_pool2_fwd(x->d, out->d, x->n, x->h, x->w, x->c, c.stream);
  if (x->needs_grad) {
    out->g_slab_ok = !x->leaf_grad && flat_slab_ok(x) && flat_slab_ok(out);
    Ctx* cp = &c;
    c.tape.push([=]() {
      Ctx& c = *cp;
      if (!out->g_set) return;
      int acc;
      float* dx = grad_acc(c, x, &acc);
      if (x->leaf_grad) {   // (its producers ran on the leaf stream: the pyramid's whole gradient chain stays there, in order)
        LeafScope leaf(c);
        launch_pool2_bwd(out->g, dx, acc, x->n, x->h, x->w, x->c, c.stream);
      } else {
        const SlabSrc ds = grad_src(out);
        launch_pool2_bwd(out->g, dx, acc, x->n, x->h, x->w, x->c, c.stream, producer_plane(c, x, true), ds.n > 1 ? &ds : nullptr);
      }
    });
  }
  return out;
}

// pool2(relu(a + b)) -- a residual block's tail (conv.py:80-85, encoder.py:63-66).  Forward-only graphs (inference, the
// post-update evaluation) run it as ONE pass that also stores the bf16 plane the next block's convolutions read; a graph that
// differentiates keeps add() and pool2()
T add_relu_pool2(Ctx& c, T a, T b) {
  static const bool on = dbg_env("EG_NO_RES_TAIL") == nullptr;
  if (on && !a->needs_grad && !b->needs_grad && a->h % 2 == 0 && a->w % 2 == 0) {
    T out = new_tensor(c, a->n, a->h / 2, a->w / 2, a->c, false);
    unsigned short* plane = producer_plane(c, out, false);
    if (launch_add_relu_pool2_fwd(a->d, b->d, out->d, plane, a->n, a->h, a->w, a->c, c.stream)) return out;
    out->d_bf = nullptr;
  }
  return pool2(c, add(c, a, b, true));
}

T add_pool2(Ctx& c, T a, T b) {
  EG_CHECK(a->n == b->n && a->h == b->h && a->w == b->w && a->c == b->c, "add_pool2 shapes");
  EG_CHECK(a->h % 2 == 0 && a->w % 2 == 0, "add_pool2 needs even extents");
  T out = new_tensor(c, a->n, a->h / 2, a->w / 2, a->c, a->needs_grad || b->needs_grad);
  if (!launch_add_pool2_fwd(a->d, b->d, out->d, a->n, a->h, a->w, a->c, c.stream)) {
    float* sum = c.arena.allocf(a->numel());
    launch_add(a->d, b->d, sum, 0, a->numel(), c.stream);
    launch_pool2_fwd(sum, out->d, a->n, a->h, a->w, a->c, c.stream);
  }
  if (out->needs_grad) {
    Ctx* cp = &c;
    c.tape.push([=]() {
      Ctx& c = *cp;
      if (!out->g_set) return;
      // d(a) = d(b) = up-sampled d(out) / 4: computed once; inputs whose gradient is still unset share the buffer
      float* shared = nullptr;
      for (T in : {a, b}) {
        if (!in->needs_grad) continue;
        if (!in->g_set && !in->g) {
          if (!shared) {
            shared = c.arena.allocf(in->numel());
            launch_pool2_bwd(out->g, shared, 0, in->n, in->h, in->w, in->c, c.stream);
          }
          in->g = shared;
          in->g_set = true;
        } else {
          int acc;
          float* d = grad_acc(c, in, &acc);
          launch_pool2_bwd(out->g, d, acc, in->n, in->h, in->w, in->c, c.stream);
        }
      }
    });
  }
  return out;
}

// the output tensors of pool2_add(a-shaped, b): the pooled sum and, with post_prelu in its float4 form, its PReLU (+ plane)
static void pool2_add_outputs(Ctx& c, int an, int ah, int aw, int ac, bool needs_grad, T b, Param* post_prelu, bool plane_only,
                              Ctx::PoolEpi* pe) {
  EG_CHECK(ah % 2 == 0 && aw % 2 == 0, "pool2_add needs even extents");
  EG_CHECK(an == b->n && ah == 2 * b->h && aw == 2 * b->w && ac == b->c, "pool2_add shapes");
  pe->b = b;
  pe->leak = post_prelu;
  pe->out = new_tensor(c, b->n, b->h, b->w, b->c, needs_grad || b->needs_grad);
  static const bool fuse2 = dbg_env("EG_NO_POOL_PRELU") == nullptr;
  const bool vec = ac % 4 == 0 && (size_t)an * ah * aw * ac < (1ull << 31);
  pe->out2 = (post_prelu && fuse2 && vec) ? new_tensor(c, b->n, b->h, b->w, b->c, false) : nullptr;
  pe->plane2 = pe->out2 ? producer_plane(c, pe->out2, false) : nullptr;
  static const bool plane_only_on = dbg_env("EG_NO_PLANE_ONLY") == nullptr;
  if (pe->out2) pe->out2->d_plane_only = plane_only_on && plane_only && pe->plane2 != nullptr;
  pe->done = false;
}
static T pool2_add_finish(Ctx& c, T a, T b, Param* post_prelu, T* prelu_out, const Ctx::PoolEpi& pe);

T pool2_add(Ctx& c, T a, T b, Param* post_prelu, T* prelu_out, bool plane_only) {
  Ctx::PoolEpi pe;
  pool2_add_outputs(c, a->n, a->h, a->w, a->c, a->needs_grad, b, post_prelu, plane_only, &pe);
  return pool2_add_finish(c, a, b, post_prelu, prelu_out, pe);
}

T conv2d_pool2_add(Ctx& c, T x, WRef* w, Param* bias, int k, T b, Param* post_prelu, T* prelu_out, bool plane_only) {
  // the convolution's output shape (SAME, stride 1) is x's with w->Cc channels
  Ctx::PoolEpi pe;
  const bool need = x->needs_grad || w->want_grad || (bias && bias->want_grad);
  pool2_add_outputs(c, x->n, x->h, x->w, w->Cc, need, b, post_prelu, plane_only, &pe);
  c.pool_epi = pe;
  c.feeds_in = FEED_FLAT;   // (a K-split launch cannot pool: its slabs go to the pool2_add pass)
  T y = conv2d(c, x, w, bias, k, 1, PAD_SAME);
  y->g_up_ok = true;   // (y's only reader is the pool below, its producer's backward handles the pooled form of its gradient)
  pe = c.pool_epi;
  c.pool_epi = Ctx::PoolEpi();
  return pool2_add_finish(c, y, b, post_prelu, prelu_out, pe);
}

// forward pass (unless the producing convolution's epilogue already did it: pe.done) + the backward of out = pool2(a) + b
static T pool2_add_finish(Ctx& c, T a, T b, Param* post_prelu, T* prelu_out, const Ctx::PoolEpi& pe) {
  T out = pe.out, out2 = pe.out2;
  // one pass where the float4 form applies: `a` may still be K-split slabs (conv2d with FEED_FLAT) and the PReLU of the
  // result -- the next unit's prelu(h) -- is a second store
  if (!pe.done) {
    const bool vec = a->c % 4 == 0 && a->numel() < (1ull << 31);
    const SlabSrc as = data_src(a);
    EG_CHECK(as.n <= 1 || vec, "pool2_add: slab operand without the float4 form");
    if (!launch_pool2_add_fwd(a->d, b->d, out->d, a->n, a->h, a->w, a->c, c.stream, as.n > 1 ? &as : nullptr,
                              out2 ? post_prelu->w : nullptr, out2 && !out2->d_plane_only ? out2->d : nullptr, pe.plane2)) {
      launch_pool2_fwd(a->d, out->d, a->n, a->h, a->w, a->c, c.stream);
      launch_axpy(b->d, out->d, 1.f, 1, out->numel(), c.stream);
    }
  }
  a->nslab = 0;   // (consumed; its own backward needs only its gradient)
  if (out->needs_grad) {
    Ctx* cp = &c;
    c.tape.push([=]() {
      Ctx& c = *cp;
      if (!out->g_set) return;
      if (b->needs_grad) {
        if (!b->g_set && !b->g && !b->g_inplace) {   // d(b) = d(out): the one buffer serves both
          b->g = out->g;
          b->g_set = true;
        } else {
          int acc;
          float* d = grad_acc(c, b, &acc);
          launch_axpy(out->g, d, 1.f, acc, b->numel(), c.stream);
        }
      }
      if (a->needs_grad) {
        static const bool colsum_small = dbg_env("EG_NO_POOLED_COLSUM") == nullptr;
        static const bool up_on = dbg_env("EG_NO_POOL_UP") == nullptr;
        if (up_on && colsum_small && a->g_up_ok && !a->g_set && (c.cfg.x3 || c.cfg.bf16) && a->h % 2 == 0 && a->w % 2 == 0) {
          // d(a) = 0.25 x the upsampling of d(out) is not written: the convolution behind a reads d(out) in both of its gradient
          // kernels (Tensor::g_up) and takes its bias gradient from d(out)'s column sums
          materialize_grad(c, out);
          a->g_up = out;
          a->g_set = true;
          a->g_bf = nullptr;
          a->g_colpart = nullptr;
          a->g_colsum_src = out->g;
          a->g_colsum_rows = out->pixels();
          return;
        }
        int acc;
        float* d = grad_acc(c, a, &acc);
        launch_pool2_bwd(out->g, d, acc, a->n, a->h, a->w, a->c, c.stream, producer_plane(c, a, true));
        if (!acc && colsum_small) {   // the column sums of d are those of out->g (each value appears 4 times, scaled by 1/4)
          a->g_colsum_src = out->g;
          a->g_colsum_rows = out->pixels();
        }
      }
    });
  }
  if (post_prelu) *prelu_out = out2 ? prelu_tail(c, out, out2, post_prelu, 0.f) : prelu(c, out, post_prelu, 0.f);
  return out;
}

T gmean(Ctx& c, T x, bool pre_relu) {
  T out = new_tensor(c, x->n, 1, 1, x->c, x->needs_grad);
  const int HW = x->h * x->w;
  launch_gmean_fwd(x->d, out->d, x->n, HW, x->c, pre_relu, c.stream);
  if (x->needs_grad) {
    Ctx* cp = &c;
    c.tape.push([=]() {
      Ctx& c = *cp;
      if (!out->g_set) return;
      int acc;
      float* dx = grad_acc(c, x, &acc);
      launch_gmean_bwd(x->d, out->g, dx, acc, x->n, HW, x->c, pre_relu, c.stream);
    });
  }
  return out;
}

// ---------------------------------------------------------------------------------------------
// tf.image.resize_images(method=2) in TF 1.14 = legacy bicubic (edgegan.py:211-213; Appendix B.3)
// ---------------------------------------------------------------------------------------------
static const ResizeTab& resize_tab(Ctx& c, int n_in, int n_out) {
  auto key = std::make_pair(n_in, n_out);
  auto it = c.resize_tabs.find(key);
  if (it != c.resize_tabs.end()) return it->second;
  // coefficient table exactly as the TF kernel builds it (1024 entries, A = -0.75, float)
  const double A = -0.75;
  std::vector<float> tab(1025 * 2);
  for (int i = 0; i <= 1024; ++i) {
    double x = i / 1024.0;
    tab[i * 2] = (float)(((A + 2) * x - (A + 3)) * x * x + 1);
    x += 1.0;
    tab[i * 2 + 1] = (float)(((A * x - 5 * A) * x + 8 * A) * x - 4 * A);
  }
  std::vector<int> fi((size_t)n_out * 4);
  std::vector<float> fw((size_t)n_out * 4);
  std::vector<std::vector<std::pair<int, float>>> back(n_in);
  const double scale = (double)n_in / n_out;
  for (int o = 0; o < n_out; ++o) {
    const double loc = scale * o;
    const int fl = (int)std::floor(loc);
    const int off = (int)std::lrint((loc - fl) * 1024);
    const float wt[4] = {tab[off * 2 + 1], tab[off * 2], tab[(1024 - off) * 2], tab[(1024 - off) * 2 + 1]};
    for (int t = 0; t < 4; ++t) {
      int idx = fl - 1 + t;
      idx = idx < 0 ? 0 : (idx > n_in - 1 ? n_in - 1 : idx);
      fi[o * 4 + t] = idx;
      fw[o * 4 + t] = wt[t];
      if (wt[t] != 0.f) back[idx].push_back({o, wt[t]});
    }
  }
  size_t bn = 1;
  for (auto& v : back) bn = v.size() > bn ? v.size() : bn;
  std::vector<int> bi((size_t)n_in * bn, 0);
  std::vector<float> bw((size_t)n_in * bn, 0.f);
  for (int i = 0; i < n_in; ++i)
    for (size_t j = 0; j < back[i].size(); ++j) {
      bi[i * bn + j] = back[i][j].first;
      bw[i * bn + j] = back[i][j].second;
    }
  ResizeTab r;
  r.n_in = n_in;
  r.n_out = n_out;
  r.fn = 4;
  r.bn = (int)bn;
  EG_HIP(hipMalloc(&r.f_i, fi.size() * 4));
  EG_HIP(hipMalloc(&r.f_w, fw.size() * 4));
  EG_HIP(hipMalloc(&r.b_i, bi.size() * 4));
  EG_HIP(hipMalloc(&r.b_w, bw.size() * 4));
  EG_HIP(hipMemcpy(r.f_i, fi.data(), fi.size() * 4, hipMemcpyHostToDevice));
  EG_HIP(hipMemcpy(r.f_w, fw.data(), fw.size() * 4, hipMemcpyHostToDevice));
  EG_HIP(hipMemcpy(r.b_i, bi.data(), bi.size() * 4, hipMemcpyHostToDevice));
  EG_HIP(hipMemcpy(r.b_w, bw.data(), bw.size() * 4, hipMemcpyHostToDevice));
  return c.resize_tabs[key] = r;
}

T resize(Ctx& c, T x, int size) {
  if (x->h == size && x->w == size) return x;
  const ResizeTab& ty = resize_tab(c, x->h, size);
  const ResizeTab& tx = resize_tab(c, x->w, size);
  T out = new_tensor(c, x->n, size, size, x->c, x->needs_grad);
  launch_resize(x->d, out->d, 0, x->n, x->h, x->w, size, size, x->c, ty.f_i, ty.f_w, ty.fn, tx.f_i, tx.f_w, tx.fn,
                c.stream);
  if (x->needs_grad) {
    Ctx* cp = &c;
    const ResizeTab* pty = &ty;
    const ResizeTab* ptx = &tx;
    c.tape.push([=]() {
      Ctx& c = *cp;
      if (!out->g_set) return;
      int acc;
      float* dx = grad_acc(c, x, &acc);
      launch_resize(out->g, dx, acc, x->n, size, size, x->h, x->w, x->c, pty->b_i, pty->b_w, pty->bn, ptx->b_i,
                    ptx->b_w, ptx->bn, c.stream);
    });
  }
  return out;
}

// tf.concat([a, b], 2) (edgegan.py:245)
T concat_w(Ctx& c, T a, T b) {
  T out = new_tensor(c, a->n, a->h, a->w + b->w, a->c, a->needs_grad || b->needs_grad);
  launch_copy_cols(a->d, out->d, a->n, a->h, a->w, out->w, 0, a->c, 1, 0, c.stream);
  launch_copy_cols(b->d, out->d, b->n, b->h, b->w, out->w, a->w, b->c, 1, 0, c.stream);
  if (out->needs_grad) {
    // one tape entry per input (a's, then b's: two consecutive indices), so that a caller which schedules the two
    // producers' backward chains separately (model.cpp, sub-steps 5 / 7) can run each input's part on its own
    Ctx* cp = &c;
    int wo = 0;
    for (T in : {a, b}) {
      c.tape.push([=]() {
        Ctx& c = *cp;
        if (!out->g_set || !in->needs_grad) return;
        int acc;
        float* d = grad_acc(c, in, &acc);
        launch_copy_cols(d, out->g, in->n, in->h, in->w, out->w, wo, in->c, 0, acc, c.stream);
      });
      wo += in->w;
    }
  }
  return out;
}

T slice_w(Ctx& c, T x, int w0, int w1) {
  T out = new_tensor(c, x->n, x->h, w1 - w0, x->c, false);
  launch_copy_cols(out->d, x->d, x->n, x->h, w1 - w0, x->w, w0, x->c, 0, 0, c.stream);
  return out;
}

// activation.py:23-32
T prelu(Ctx& c, T x, Param* leak, float leak_const) {
  T out = new_tensor(c, x->n, x->h, x->w, x->c, false);
  launch_prelu_fwd(x->d, leak ? leak->w : nullptr, leak_const, out->d, x->numel(), c.stream, producer_plane(c, out, false));
  return prelu_tail(c, x, out, leak, leak_const);
}
// the backward of out = prelu(x) (whoever computed out: prelu() above, a convolution's epilogue, pool2_add's second store)
static T prelu_tail(Ctx& c, T x, T out, Param* leak, float leak_const) {
  const bool need = x->needs_grad || (leak && leak->want_grad);
  if (need) {
    out->needs_grad = true;
    out->g_slab_ok = flat_slab_ok(out) && (x->numel() & 3) == 0;
    if (leak && x->needs_grad) {   // (conv_dgrad may take the backward into its epilogue)
      out->prelu_pre = x;
      out->prelu_leak = leak;
    }
    Ctx* cp = &c;
    c.tape.push([=]() {
      Ctx& c = *cp;
      if (!out->g_set) return;
      if (out->g_act_done) {   // the convolution that wrote out's gradient applied this backward: the buffer holds d(x)
        // (conv_dgrad took the epilogue only because x had no gradient and no other writer yet; a consumer of x recorded between
        //  this op and that convolution would have written x->g since -- refuse loudly, as grad_acc does, instead of dropping it)
        EG_CHECK(!x->g_set && (x->g == nullptr || x->g == out->g), "prelu backward in the epilogue: the PReLU's input already holds a gradient");
        x->g = out->g;
        x->g_set = true;
        x->g_bf = out->g_bf;
        x->g_bf_stream = out->g_bf_stream;
        x->g_plane_only = out->g_plane_only;
        x->g_colsum_src = nullptr;
        x->g_colpart = nullptr;
        return;
      }
      float* dx = nullptr;
      int acc = 0;
      if (x->needs_grad) dx = grad_acc(c, x, &acc);
      const bool pg = leak && leak->want_grad && c.param_grad_enabled;
      if (!dx && !pg) return;
      need_f32(c, x);   // (a pre-activation kept as its plane alone: this pass reads its values)
      const SlabSrc ds = grad_src(out);
      const int prow = dx ? flat_colpart_rows(x->numel(), x->c) : 0;
      float* part = dx ? colpart_for(c, x, acc, prow) : nullptr;
      unsigned short* gplane = dx ? producer_plane(c, x, true) : nullptr;
      launch_prelu_bwd(x->d, out->g, leak ? leak->w : nullptr, leak_const, dx, dx ? plane_only_acc(c, x, acc, gplane) : acc,
                       pg ? leak->g : nullptr, pg && leak->g_set ? 1 : 0, x->numel(), c.scratch, c.stream, gplane,
                       ds.n > 1 ? &ds : nullptr, part, x->c);
      colpart_done(x, part, prow);
      if (pg) leak->g_set = true;
    });
  }
  return out;
}

// conv.py:197-198
T minmax(Ctx& c, T x, float pre_leak) {
  T out = new_tensor(c, x->n, x->h, x->w, x->c, x->needs_grad);
  T mm = new_tensor(c, x->n, 1, 1, 2 * x->c, false);
  const int HW = x->h * x->w;
  const SlabSrc xs = data_src(x);   // (conv2d with FEED_MINMAX: the pass sums the slabs and stores the sum to x->d)
  auto mm_scratch = [](Ctx& c, int NB, int HW, int C) { return minmax_chunked_shape(NB, HW, C) ? c.arena.allocf(minmax_chunk_scratch_floats(NB, HW, C)) : nullptr; };
  launch_minmax_fwd(x->d, out->d, mm->d, x->n, HW, x->c, c.stream, pre_leak, xs.n > 1 ? &xs : nullptr, x->d, mm_scratch(c, x->n, HW, x->c));
  x->nslab = 0;
  if (x->needs_grad) {
    Ctx* cp = &c;
    c.tape.push([=]() {
      Ctx& c = *cp;
      if (!out->g_set) return;
      int acc;
      float* dx = grad_acc(c, x, &acc);
      const int prow = minmax_bwd_colpart_rows(x->n, HW, x->c);
      float* part = colpart_for(c, x, acc, prow);
      unsigned short* gplane = producer_plane(c, x, true);
      launch_minmax_bwd(x->d, mm->d, out->g, dx, plane_only_acc(c, x, acc, gplane), x->n, HW, x->c, c.stream, pre_leak, gplane, part,
                        mm_scratch(c, x->n, HW, x->c));
      colpart_done(x, part, prow);
    });
  }
  return out;
}

// conv.py:207: ht_plus = ht + rg * img_new
T gate(Ctx& c, T h, T rg, T img) {
  T out = new_tensor(c, h->n, h->h, h->w, h->c, h->needs_grad || rg->needs_grad || img->needs_grad);
  launch_gate_fwd(h->d, rg->d, img->d, out->d, h->numel(), c.stream);
  if (out->needs_grad) {
    Ctx* cp = &c;
    c.tape.push([=]() {
      Ctx& c = *cp;
      if (!out->g_set) return;
      float* dh = nullptr;
      int acc_h = 0, a1, a2;
      if (h->needs_grad) dh = grad_acc(c, h, &acc_h);
      float* drg = grad_acc(c, rg, &a1);
      float* dimg = grad_acc(c, img, &a2);
      EG_CHECK(!a1 && !a2, "gate operands are single-use");
      launch_gate_bwd(rg->d, img->d, out->g, dh, acc_h, drg, dimg, h->numel(), c.stream);
    });
  }
  return out;
}

// conv.py:207-209 in one pass: prelu(ht + rg * img_new, leak).  ht_plus is not materialised: the backward recomputes it
// from its three inputs for the PReLU mask and the slope's gradient (exactly the values the two-pass form reads back)
T gate_prelu(Ctx& c, T h, T rg, T img, Param* leak, bool plane_only) {
  T out = new_tensor(c, h->n, h->h, h->w, h->c, true);
  unsigned short* plane = producer_plane(c, out, false);
  static const bool plane_only_on = dbg_env("EG_NO_PLANE_ONLY") == nullptr;
  out->d_plane_only = plane_only_on && plane_only && plane != nullptr && (h->numel() & 3) == 0;
  launch_gate_prelu_fwd(h->d, rg->d, img->d, leak->w, out->d_plane_only ? nullptr : out->d, h->numel(), c.stream, plane);
  out->g_slab_ok = flat_slab_ok(out);
  Ctx* cp = &c;
  c.tape.push([=]() {
    Ctx& c = *cp;
    if (!out->g_set) return;
    float* dh = nullptr;
    int acc_h = 0, a1, a2;
    if (h->needs_grad) dh = grad_acc(c, h, &acc_h);
    float* drg = grad_acc(c, rg, &a1);
    float* dimg = grad_acc(c, img, &a2);
    EG_CHECK(!a1 && !a2, "gate operands are single-use");
    const bool pg = leak->want_grad && c.param_grad_enabled;
    const SlabSrc ds = grad_src(out);
    const int prow = flat_colpart_rows(img->numel(), img->c);
    float* part = colpart_for(c, img, a2, prow);     // the image convolution's bias gradient: column sums of dimg
    launch_gate_prelu_bwd(h->d, rg->d, img->d, leak->w, out->g, dh, acc_h, drg, dimg, pg ? leak->g : nullptr,
                          pg && leak->g_set ? 1 : 0, h->numel(), c.scratch, c.stream, ds.n > 1 ? &ds : nullptr, part, img->c);
    colpart_done(img, part, prow);
    if (pg) leak->g_set = true;
  });
  return out;
}

// encoder.py:78-79
T reparam(Ctx& c, T mu, T ls, const float* eps_dev) {
  T z = new_tensor(c, mu->n, 1, 1, mu->c, mu->needs_grad);
  launch_reparam_fwd(mu->d, ls->d, eps_dev, z->d, z->numel(), c.stream);
  if (z->needs_grad) {
    Ctx* cp = &c;
    c.tape.push([=]() {
      Ctx& c = *cp;
      if (!z->g_set) return;
      int a1, a2;
      float* dmu = grad_acc(c, mu, &a1);
      float* dls = grad_acc(c, ls, &a2);
      EG_CHECK(!a1 && !a2, "reparam operands are single-use");
      launch_reparam_bwd(ls->d, eps_dev, z->g, dmu, dls, z->numel(), c.stream);
    });
  }
  return z;
}

// normalization.py:38-76 for every live SN weight of the classifier, one batched launch sequence
void sn_prepare(Ctx& c) {
  const int count = (int)c.sn.size();
  if (count == 0) return;
  if (!c.sn_valid) {  // the classifier's weights only change in sub-step 4: sub-steps 5 and 7 share one W/sigma
    launch_sn_fwd(c.sn_table, count, c.sn_maxK, c.sn_maxCo, c.stream);
    for (auto& s : c.sn) {
      s.wT_valid = s.wL[0].valid = s.wL[1].valid = false;
      if (!s.split_rows) continue;
      // the update gate's filter by input-channel rows (hidden state | image): copies that live as long as Wbar does
      const int ct = s.split_rows + 3;
      launch_copy_cols(s.part[0], s.Wbar, s.T, 1, s.split_rows, ct, 0, s.Co, 0, 0, c.stream);
      launch_copy_cols(s.part[1], s.Wbar, s.T, 1, 3, ct, s.split_rows, s.Co, 0, 0, c.stream);
      for (int i = 0; i < 2; ++i) s.partT_valid[i] = s.partL[i][0].valid = s.partL[i][1].valid = false;
    }
    c.sn_valid = true;
  }
  bool want = false;
  for (auto& s : c.sn) want |= s.W->want_grad;
  if (want) {
    Ctx* cp = &c;
    c.tape.push([=]() {
      Ctx& c = *cp;
      if (!c.param_grad_enabled) return;
      for (auto& s : c.sn) EG_CHECK(s.dWbar_set, "SN weight %s received no gradient", s.W->name.c_str());
      flush_deferred(c);   // dWbar of every SN weight is complete before sigma's backward reads it
      launch_sn_bwd(c.sn_table, (int)c.sn.size(), c.sn_maxK, c.sn_maxCo, c.stream);
      for (auto& s : c.sn) s.W->g_set = true;
    });
  }
}

Ctx::~Ctx() {
  for (auto& kv : resize_tabs) {
    (void)hipFree(kv.second.f_i);
    (void)hipFree(kv.second.f_w);
    (void)hipFree(kv.second.b_i);
    (void)hipFree(kv.second.b_w);
  }
  for (auto e : ev_pool) (void)hipEventDestroy(e);
  for (auto& r : prof) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  for (auto& p : params)
    {
      if (p->wT) (void)hipFree(p->wT);
      if (p->wM) (void)hipFree(p->wM);
      for (int o = 0; o < 2; ++o) p->wL[o].release();
    }
  for (int i = 0; i < 2; ++i) {
    if (feed_host[i]) (void)hipHostFree(feed_host[i]);
    if (loss_host[i]) (void)hipHostFree(loss_host[i]);
    if (feed_copied[i]) (void)hipEventDestroy(feed_copied[i]);
    if (loss_ready[i]) (void)hipEventDestroy(loss_ready[i]);
  }
  (void)hipFree(inf_in);
  for (int i = 0; i < kInferSlots; ++i) {
    (void)hipFree(inf_edge[i]);
    (void)hipFree(inf_image[i]);
  }
  for (auto e : ev_infer)
    if (e) (void)hipEventDestroy(e);
  (void)hipFree(inf_eps);
  (void)hipFree(inf_cls);
  groups[G_G2].g = nullptr;   // lives inside G1's allocation (model.cpp: build_context)
  for (auto& g : groups) {
    (void)hipFree(g.w);
    (void)hipFree(g.g);
    (void)hipFree(g.ms);
    (void)hipFree(g.mom);
  }
  for (auto& s : sn) {
    (void)hipFree(s.Wbar);
    (void)hipFree(s.WbarT);
    for (int o = 0; o < 2; ++o) s.wL[o].release();
    for (int i = 0; i < 2; ++i) {
      (void)hipFree(s.part[i]);
      (void)hipFree(s.partT[i]);
      for (int o = 0; o < 2; ++o) s.partL[i][o].release();
    }
    (void)hipFree(s.dWbar);
    (void)hipFree(s.t);
    (void)hipFree(s.sv);
    (void)hipFree(s.dv);
    (void)hipFree(s.scal);
  }
  (void)hipFree(sn_table);
  (void)hipFree(lanes[0].scratch ? lanes[0].scratch : scratch);
  (void)hipFree(d_images);
  (void)hipFree(d_z);
  (void)hipFree(d_alpha);
  (void)hipFree(d_eps);
  (void)hipFree(d_losses);
  (void)hipFree(d_eval);
  (void)hipFree(d_classes);
  for (int i = 0; i < 2; ++i) {
    (void)hipFree(d_gen_cache[i]);
    (void)hipFree(d_gen_bn[i]);
    for (int j = 0; j < 3; ++j) (void)hipFree(d_gen_bn_blk[i][j]);
  }
  // the context's own stream / scratch are lane 0's; make them the active ones again before freeing
  if (lanes[0].stream) stream = lanes[0].stream;
  for (int i = 0; i < kLanes; ++i) {
    if (i > 0) {
      if (lanes[i].stream) (void)hipStreamDestroy(lanes[i].stream);
      if (lanes[i].scratch) (void)hipFree(lanes[i].scratch);
    }
    if (lanes[i].done) (void)hipEventDestroy(lanes[i].done);
    if (lanes[i].leaf) (void)hipStreamDestroy(lanes[i].leaf);
    if (lanes[i].leaf_in) (void)hipEventDestroy(lanes[i].leaf_in);
    if (lanes[i].leaf_out) (void)hipEventDestroy(lanes[i].leaf_out);
  }
  for (auto st : placeholder_streams) (void)hipStreamDestroy(st);
  if (ev_fork) (void)hipEventDestroy(ev_fork);
  if (ev_gen) (void)hipEventDestroy(ev_gen);
  if (ev_g1fwd) (void)hipEventDestroy(ev_g1fwd);
  if (ev_branch) (void)hipEventDestroy(ev_branch);
  if (ev_cls) (void)hipEventDestroy(ev_cls);
  if (stream) (void)hipStreamDestroy(stream);
}

}  // namespace eg
