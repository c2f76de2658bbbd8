// Variable registry, the seven networks' forward graphs and the seven optimizer sub-steps of
// EdgeGAN.update_model (edgegan/models/edgegan.py:87-130, 202-342), on the tape ops of ops.cpp.
#include <cmath>
#include <cstdlib>

#include "engine.h"

namespace eg {

static const char* kGroupPrefix[G_COUNT] = {"G1", "G2", "D2", "D", "D_patch2", "D_patch3", "E"};
static inline int ceil_half(int n) { return (n + 1) / 2; }

// ---------------------------------------------------------------------------------------------
// registry: names / shapes / initialisers, creation order of
// edgegan.py:202-275: G1, G2, D2, D, D_patch2, D_patch3, E
// ---------------------------------------------------------------------------------------------
// the variables contrib batch_norm creates under <scope>/batch_norm (normalization.py:20-25; creation order beta, gamma, moving_*)
static void bn_vars(std::vector<VarInfo>& v, const std::string& scope, int C, int group) {
  v.push_back({scope + "/batch_norm/BatchNorm/beta", {C}, true, false, 0, 0.f, group});
  v.push_back({scope + "/batch_norm/BatchNorm/gamma", {C}, true, false, 0, 1.f, group});
  v.push_back({scope + "/batch_norm/BatchNorm/moving_mean", {C}, false, false, 0, 0.f, group});
  v.push_back({scope + "/batch_norm/BatchNorm/moving_variance", {C}, false, false, 0, 1.f, group});
}

static void gen_vars(std::vector<VarInfo>& v, const Config& cfg, int group) {
  const std::string p = kGroupPrefix[group];
  int h = cfg.H, w = cfg.W / 2;
  for (int i = 0; i < 4; ++i) {
    h = ceil_half(h);
    w = ceil_half(w);
  }
  const int gf = cfg.gf;
  v.push_back({p + "/g_lin_0/Matrix", {cfg.g_in(), gf * 8 * h * w}, true, false, 1, 0.02f, group});
  v.push_back({p + "/g_lin_0/bias", {gf * 8 * h * w}, true, false, 0, 0.f, group});
  v.push_back({p + "/batch_norm/BatchNorm/beta", {gf * 8}, true, false, 0, 0.f, group});
  v.push_back({p + "/batch_norm/BatchNorm/gamma", {gf * 8}, true, false, 0, 1.f, group});
  v.push_back({p + "/batch_norm/BatchNorm/moving_mean", {gf * 8}, false, false, 0, 0.f, group});
  v.push_back({p + "/batch_norm/BatchNorm/moving_variance", {gf * 8}, false, false, 0, 1.f, group});
  const int ch[5] = {gf * 8, gf * 4, gf * 2, gf, cfg.c_dim};
  for (int i = 1; i <= 4; ++i) {
    const std::string b = p + "/g_dconv_" + std::to_string(i) + "/deconv2d/";
    v.push_back({b + "w", {5, 5, ch[i], ch[i - 1]}, true, false, 1, 0.02f, group});
    v.push_back({b + "b", {ch[i]}, true, false, 0, 0.f, group});
    if (cfg.g_bn && i < 4) bn_vars(v, p + "/g_dconv_" + std::to_string(i), ch[i], group);   // deconv_block (conv.py:124-130), G_norm = 'batch'
  }
}

static const char* kDiscNames[4] = {"d_conv_0", "d_conv_1", "d_conv_3", "d_conv_4"};

static void disc_vars(std::vector<VarInfo>& v, const Config& cfg, int group, int in_h, int in_w) {
  const std::string p = kGroupPrefix[group];
  const int df = cfg.df;
  const int ch[5] = {cfg.c_dim, df, df * 2, df * 4, df * 8};
  int h = in_h, w = in_w;
  for (int i = 0; i < 4; ++i) {
    v.push_back({p + "/" + kDiscNames[i] + "/conv2d/w", {4, 4, ch[i], ch[i + 1]}, true, false, 2, 0.02f, group});
    if (cfg.d_bn && i > 0) bn_vars(v, p + "/" + kDiscNames[i], ch[i + 1], group);   // conv_block (conv.py:61-67), D_norm = 'batch'
    h = ceil_half(h);
    w = ceil_half(w);
  }
  v.push_back({p + "/d_linear_5/Matrix", {h * w * ch[4], 1}, true, false, 1, 0.02f, group});
  v.push_back({p + "/d_linear_5/bias", {1}, true, false, 0, 0.f, group});
}

static std::vector<int> encoder_filters(const Config& cfg) {
  std::vector<int> nf = {128, 256, 512, 512};
  if (cfg.H == 256) nf.push_back(512);  // encoder.py:57-58
  return nf;
}

static void enc_vars(std::vector<VarInfo>& v, const Config& cfg) {
  v.push_back({"E/e_resnet_64_0/conv2d/w", {4, 4, cfg.c_dim, 64}, true, false, 2, 0.02f, G_E});
  v.push_back({"E/e_resnet_64_0/conv2d/b", {64}, true, false, 0, 0.f, G_E});
  int cin = 64, i = 1;
  for (int n : encoder_filters(cfg)) {
    const std::string b = "E/e_resnet_" + std::to_string(n) + "_" + std::to_string(i++);
    v.push_back({b + "/res1/conv2d/w", {3, 3, cin, n}, true, false, 2, 0.02f, G_E});
    v.push_back({b + "/res1/conv2d/b", {n}, true, false, 0, 0.f, G_E});
    if (cfg.e_bn) bn_vars(v, b + "/res1", n, G_E);   // residual (conv.py:70-85), E_norm = 'batch'
    v.push_back({b + "/res2/conv2d/w", {3, 3, n, n}, true, false, 2, 0.02f, G_E});
    v.push_back({b + "/res2/conv2d/b", {n}, true, false, 0, 0.f, G_E});
    if (cfg.e_bn) bn_vars(v, b + "/res2", n, G_E);
    v.push_back({b + "/shortcut/conv2d/w", {1, 1, cin, n}, true, false, 2, 0.02f, G_E});
    v.push_back({b + "/shortcut/conv2d/b", {n}, true, false, 0, 0.f, G_E});
    cin = n;
  }
  for (const char* head : {"FC8_mu", "FC8_sigma"}) {
    v.push_back({std::string("E/") + head + "/w", {cin, cfg.z_dim}, true, false, 1, 0.02f, G_E});
    v.push_back({std::string("E/") + head + "/b", {cfg.z_dim}, true, false, 0, 0.f, G_E});
  }
}

static std::string sn_u_name(const std::string& scope) { return scope + "/" + scope + "/u"; }

static void cls_conv_vars(std::vector<VarInfo>& v, const std::string& scope, int k, int cin, int cout,
                          float bias_init, bool prelu_param, bool dead) {
  v.push_back({scope + "/weights", {k, k, cin, cout}, true, dead, 1, 0.02f, G_D2});
  v.push_back({sn_u_name(scope), {1, cout}, false, false, 2, 1.0f, G_D2});
  v.push_back({scope + "/biases", {1, cout, 1, 1}, true, dead, 0, bias_init, G_D2});
  if (prelu_param) v.push_back({scope + "/prelu/param", {}, true, false, 0, 0.2f, G_D2});
}

static const int kMruDepth[4] = {128, 256, 512, 768};

static void cls_vars(std::vector<VarInfo>& v, const Config& cfg) {
  cls_conv_vars(v, "D2/Conv", 7, 3, 8, 0.f, true, false);
  int hidden = 8;
  for (int u = 1; u <= 4; ++u) {
    const std::string b = "D2/mru_conv_unit_t_" + std::to_string(u) + "_layer_0";
    const int depth = kMruDepth[u - 1];
    v.push_back({b + "/norm_activation_in/prelu/param", {}, true, false, 0, 0.2f, G_D2});
    cls_conv_vars(v, b + "/update_gate", 3, hidden + 3, hidden, 0.5f, false, false);
    cls_conv_vars(v, b + "/Conv", 3, 3, hidden, 0.f, false, false);
    v.push_back({b + "/norm_activation_merge_1/prelu/param", {}, true, false, 0, 0.2f, G_D2});
    cls_conv_vars(v, b + "/Conv_1", 3, hidden, depth, 0.f, true, false);
    cls_conv_vars(v, b + "/Conv_2", 3, depth, depth, 0.f, false, false);
    cls_conv_vars(v, b + "/Conv_3", 1, hidden, depth, 0.f, false, false);
    hidden = depth;
  }
  v.push_back({"D2/mru_conv_unit_last_norm/prelu/param", {}, true, false, 0, 0.2f, G_D2});
  cls_conv_vars(v, "D2/Conv_1", 1, hidden, 1, 0.f, false, true);  // dead `disc` head (quirk N8)
  v.push_back({"D2/fully_connected/weights", {hidden, cfg.num_classes}, true, false, 3, 0.f, G_D2});
  v.push_back({sn_u_name("D2/fully_connected"), {1, cfg.num_classes}, false, false, 2, 1.0f, G_D2});
  v.push_back({"D2/fully_connected/biases", {cfg.num_classes}, true, false, 0, 0.f, G_D2});
}

std::vector<VarInfo> enumerate_variables(const Config& cfg) {
  std::vector<VarInfo> v;
  gen_vars(v, cfg, G_G1);
  gen_vars(v, cfg, G_G2);
  if (cfg.multiclasses()) cls_vars(v, cfg);
  disc_vars(v, cfg, G_D, cfg.H, cfg.W);
  // edgegan.py:164-174: a patch discriminator that is switched off is never constructed (no variables, no optimizer slots)
  if (cfg.use_image_dis) disc_vars(v, cfg, G_DP2, cfg.image_dis, cfg.image_dis);
  if (cfg.use_edge_dis) disc_vars(v, cfg, G_DP3, cfg.edge_dis, cfg.edge_dis);
  enc_vars(v, cfg);
  return v;
}

Param* P(Ctx& c, const std::string& name) {
  auto it = c.by_name.find(name);
  if (it == c.by_name.end()) fail(1, "unknown variable '%s'", name.c_str());
  return it->second;
}

void build_context(Ctx& c) {
  const Config& cfg = c.cfg;
  EG_CHECK(cfg.H % 16 == 0 && (cfg.W / 2) % 16 == 0 && cfg.W % 2 == 0, "H and W/2 must be multiples of 16");
  EG_CHECK(cfg.B >= 1 && cfg.B <= 1024, "batch size out of range");
  EG_CHECK(cfg.num_classes <= 64, "num_classes <= 64");
  EG_CHECK((!cfg.use_image_dis || (cfg.image_dis >= 16 && cfg.image_dis <= 1024)) && (!cfg.use_edge_dis || (cfg.edge_dis >= 16 && cfg.edge_dis <= 1024)),
           "image_dis_size / edge_dis_size out of range (16..1024)");
  EG_HIP(hipSetDevice(cfg.device));
  // (the compute lanes are ordinary streams: creating them with the complement of the exchange's CU mask, EG_COMM_CUS in comm.cpp, was
  //  measured in round 6 -- 1,744 -> 1,246 / 1,236 / 1,242 images/s for 8 / 16 / 32 excluded compute units: a CU-masked stream costs
  //  the matrix kernels 29 % whatever the mask's size, profiles/r06j_cu_mask_cost.txt -- and removed)
  auto new_stream = [&](hipStream_t* s) { EG_HIP(hipStreamCreateWithFlags(s, hipStreamNonBlocking)); };
  new_stream(&c.stream);
  for (int g = 0; g < G_COUNT; ++g) c.groups[g].prefix = kGroupPrefix[g];
  for (const VarInfo& vi : enumerate_variables(cfg)) {
    auto p = std::make_unique<Param>();
    p->name = vi.name;
    p->shape = vi.shape;
    p->trainable = vi.trainable;
    p->dead = vi.dead;
    p->init_kind = vi.init_kind;
    p->init_param = vi.init_param;
    p->numel = 1;
    for (int d : vi.shape) p->numel *= (size_t)d;
    p->cache_T = true;
    c.by_name[p->name] = p.get();
    c.groups[vi.group].params.push_back(p.get());
    c.params.push_back(std::move(p));
  }
  // flat layout per group: [live trainables | dead trainables | non-trainables], 64-float aligned slots
  auto align = [](size_t n) { return (n + 63) & ~(size_t)63; };
  {  // G2's trainable extent, needed when G1's gradient buffer is allocated (same layout rule as the loop below)
    size_t n = 0;
    for (Param* p : c.groups[G_G2].params)
      if (p->trainable) n += align(p->numel);
    c.groups[G_G2].n_train_hint = n;
  }
  for (int g = 0; g < G_COUNT; ++g) {
    NetGroup& ng = c.groups[g];
    if (ng.params.empty()) continue;
    size_t off = 0;
    std::vector<std::pair<Param*, size_t>> place;
    for (int pass = 0; pass < 3; ++pass) {
      for (Param* p : ng.params) {
        const int cls = !p->trainable ? 2 : (p->dead ? 1 : 0);
        if (cls != pass) continue;
        place.push_back({p, off});
        off += align(p->numel);
      }
      if (pass == 0) ng.n_live = off;
      if (pass == 1) ng.n_train = off;
    }
    ng.n_total = off;
    EG_CHECK(g != G_G2 || ng.n_train == ng.n_train_hint, "G2 layout");
    EG_HIP(hipMalloc(&ng.w, ng.n_total * sizeof(float)));
    EG_HIP(hipMemsetAsync(ng.w, 0, ng.n_total * sizeof(float), c.stream));
    if (ng.n_train) {
      // G1's and G2's gradients live in ONE allocation (G2's right behind G1's): sub-steps 5 / 7 optimise both, and
      // under data parallelism they travel in one all-reduce (apply_generators)
      if (g == G_G1) {
        EG_HIP(hipMalloc(&ng.g, (ng.n_train + c.groups[G_G2].n_train_hint) * sizeof(float)));
      } else if (g == G_G2) {
        ng.g = c.groups[G_G1].g + c.groups[G_G1].n_train;
      } else {
        EG_HIP(hipMalloc(&ng.g, ng.n_train * sizeof(float)));
      }
      EG_HIP(hipMalloc(&ng.ms, ng.n_train * sizeof(float)));
      EG_HIP(hipMalloc(&ng.mom, ng.n_train * sizeof(float)));
      EG_HIP(hipMemsetAsync(ng.g, 0, ng.n_train * sizeof(float), c.stream));
      EG_HIP(hipMemsetAsync(ng.mom, 0, ng.n_train * sizeof(float), c.stream));
      launch_fill(ng.ms, 1.0f, ng.n_train, c.stream);  // RMSProp `rms` slot starts at ones
    }
    for (auto& pr : place) {
      Param* p = pr.first;
      p->w = ng.w + pr.second;
      if (p->trainable) {
        p->g = ng.g + pr.second;
        p->ms = ng.ms + pr.second;
        p->mom = ng.mom + pr.second;
      }
      if (p->init_kind == 0 && p->init_param != 0.f) launch_fill(p->w, p->init_param, p->numel, c.stream);
    }
  }
  // spectral-norm bookkeeping (classifier): one persistent buffer set per live SN weight
  if (cfg.multiclasses()) {
    for (Param* p : c.groups[G_D2].params) {
      const bool is_w = p->name.size() > 8 && p->name.compare(p->name.size() - 8, 8, "/weights") == 0;
      if (!is_w || p->dead) continue;
      const std::string scope = p->name.substr(0, p->name.size() - 8);
      SNWeight s;
      s.W = p;
      s.u = P(c, sn_u_name(scope));
      s.Co = p->shape.back();
      s.K = (int)(p->numel / s.Co);
      s.T = p->shape.size() == 4 ? p->shape[0] * p->shape[1] : 1;
      EG_HIP(hipMalloc(&s.Wbar, p->numel * 4));
      EG_HIP(hipMalloc(&s.WbarT, p->numel * 4));
      EG_HIP(hipMalloc(&s.dWbar, p->numel * 4));
      EG_HIP(hipMalloc(&s.t, (size_t)s.K * 4));
      EG_HIP(hipMalloc(&s.dv, (size_t)s.K * 4));
      EG_HIP(hipMalloc(&s.sv, (size_t)sn_sv_floats(s.Co) * 4));
      EG_HIP(hipMalloc(&s.scal, SN_SCAL_FLOATS * 4));
      if (scope.size() > 12 && scope.compare(scope.size() - 12, 12, "/update_gate") == 0) {   // [3,3,hidden+3,hidden]
        s.split_rows = p->shape[2] - 3;
        const size_t nh = (size_t)s.T * s.split_rows * s.Co, nx = (size_t)s.T * 3 * s.Co;
        EG_HIP(hipMalloc(&s.part[0], nh * 4));
        EG_HIP(hipMalloc(&s.partT[0], nh * 4));
        EG_HIP(hipMalloc(&s.part[1], nx * 4));
        EG_HIP(hipMalloc(&s.partT[1], nx * 4));
      }
      c.sn_index[p] = (int)c.sn.size();
      c.sn.push_back(s);
      if (s.K > c.sn_maxK) c.sn_maxK = s.K;
      if (s.Co > c.sn_maxCo) c.sn_maxCo = s.Co;
    }
    std::vector<SNDesc> tab(c.sn.size());
    for (size_t i = 0; i < c.sn.size(); ++i) {
      const SNWeight& s = c.sn[i];
      tab[i] = SNDesc{s.W->w, s.u->w, s.Wbar, s.t, s.sv, s.scal, s.dWbar, s.W->g, s.dv, s.K, s.Co, 0};
    }
    EG_HIP(hipMalloc(&c.sn_table, tab.size() * sizeof(SNDesc)));
    EG_HIP(hipMemcpy(c.sn_table, tab.data(), tab.size() * sizeof(SNDesc), hipMemcpyHostToDevice));
  }
  c.scratch_floats = (size_t)256 * 16384;
  EG_HIP(hipMalloc(&c.scratch, c.scratch_floats * 4));
  const size_t img = (size_t)cfg.B * cfg.H * cfg.W * cfg.c_dim;
  EG_HIP(hipMalloc(&c.d_images, img * 4));
  EG_HIP(hipMalloc(&c.d_z, (size_t)cfg.B * (cfg.z_dim + 1) * 4));
  EG_HIP(hipMalloc(&c.d_alpha, (size_t)3 * cfg.B * 4));
  EG_HIP(hipMalloc(&c.d_eps, 16 * 4));
  EG_HIP(hipMalloc(&c.d_losses, 64 * 4));
  EG_HIP(hipMalloc(&c.d_classes, (size_t)cfg.B * 4));
  EG_HIP(hipMalloc(&c.d_eval, ((size_t)16 + 6 * cfg.B + 3 * cfg.B + 4) * 4));   // [16 losses | 6 x B logits | alpha | eps]
  EG_HIP(hipMemsetAsync(c.d_eval, 0, ((size_t)16 + 9 * cfg.B + 4) * 4, c.stream));
  for (int i = 0; i < 2; ++i) {
    EG_HIP(hipMalloc(&c.d_gen_cache[i], img / 2 * 4));
    EG_HIP(hipMalloc(&c.d_gen_bn[i], (size_t)2 * cfg.gf * 8 * 4));
    for (int j = 0; cfg.g_bn && j < 3; ++j) EG_HIP(hipMalloc(&c.d_gen_bn_blk[i][j], (size_t)2 * (cfg.gf * 4 >> j) * 4));
  }
  EG_HIP(hipMemsetAsync(c.d_losses, 0, 64 * 4, c.stream));
  EG_HIP(hipMemsetAsync(c.d_eps, 0, 16 * 4, c.stream));
  EG_HIP(hipStreamSynchronize(c.stream));
  // lane 0 is the context's own stream / scratch; lanes 1.. get their own
  c.lanes[0].stream = c.stream;
  c.lanes[0].scratch = c.scratch;
  // Stream -> hardware queue placement.  The HIP runtime deals a process's streams onto GPU_MAX_HW_QUEUES (4) hardware
  // queues in CREATION order, round-robin, and two streams on one queue serialise; more queues are not an option (8 / 16
  // queues measured -19 % / -25 %: the big one-block-per-CU matrix kernels of different streams then really share the
  // chip; profiles/r03q_hw_queues.txt).  Which streams share a queue is worth up to 13 % (profiles/r03r_stream_plan_sweep*.txt,
  // 14 placements x 4 workloads): best everywhere is  lane 0 | lane 1 + ALL leaf streams | lane 2 | lane 3  -- the four
  // dependent chains never wait for one another, and the filter gradients (leaves) queue up behind one another and behind the
  // image-discriminator lane.  The plan string is the creation order: a-d = the lanes' main streams (a = the context's own
  // stream, created first in build_context), A-D their leaf streams, x = a placeholder stream that only advances the
  // round-robin.  EG_STREAM_PLAN (experiments build) overrides it for A/B runs.  (The placement assumes the engine is the
  // first creator of streams in the process, as it is under bench.py / train.py / test.py.)
  {
    const char* plan = dbg_env("EG_STREAM_PLAN") ? dbg_env("EG_STREAM_PLAN") : "bcdxAxxxBxxxCxxxD";
    for (const char* q = plan; *q; ++q) {
      hipStream_t* slot = nullptr;
      hipStream_t dummy = nullptr;
      if (*q >= 'b' && *q <= 'd') slot = &c.lanes[*q - 'a'].stream;
      else if (*q >= 'A' && *q <= 'D') slot = &c.lanes[*q - 'A'].leaf;
      else if (*q == 'x') slot = &dummy;
      EG_CHECK(slot && !*slot, "EG_STREAM_PLAN: bad or repeated token '%c'", *q);
      if (slot == &dummy) EG_HIP(hipStreamCreateWithFlags(slot, hipStreamNonBlocking));
      else new_stream(slot);
      if (slot == &dummy) c.placeholder_streams.push_back(dummy);
    }
    for (int i = 0; i < Ctx::kTrainLanes; ++i) EG_CHECK(c.lanes[i].stream && c.lanes[i].leaf, "EG_STREAM_PLAN misses a stream");
  }
  for (int i = 0; i < Ctx::kTrainLanes; ++i) {
    if (i > 0) EG_HIP(hipMalloc(&c.lanes[i].scratch, c.scratch_floats * 4));
    EG_HIP(hipEventCreateWithFlags(&c.lanes[i].done, hipEventDisableTiming));
    EG_HIP(hipEventCreateWithFlags(&c.lanes[i].leaf_in, hipEventDisableTiming));
    EG_HIP(hipEventCreateWithFlags(&c.lanes[i].leaf_out, hipEventDisableTiming));
  }
  EG_HIP(hipEventCreateWithFlags(&c.ev_fork, hipEventDisableTiming));
  EG_HIP(hipEventCreateWithFlags(&c.ev_gen, hipEventDisableTiming));
  EG_HIP(hipEventCreateWithFlags(&c.ev_g1fwd, hipEventDisableTiming));
  EG_HIP(hipEventCreateWithFlags(&c.ev_branch, hipEventDisableTiming));
  EG_HIP(hipEventCreateWithFlags(&c.ev_cls, hipEventDisableTiming));
  for (int i = 0; i < Ctx::kLanes; ++i) EG_HIP(hipEventCreateWithFlags(&c.ev_infer[i], hipEventDisableTiming));
  if (const char* e = getenv("EG_INFER_SLOTS")) {   // calls in flight: 4 (one chain each), 2 (two lanes each), 1 (one call, two lanes: round 5)
    const int v = atoi(e);
    if (v == 1 || v == 2 || v == 4 || v == 8) c.inf_slots = v;
  }
  if (const char* e = getenv("EG_LANES")) {   // 0 = one stream; bit 0 = sub-steps 1-4 concurrent, bit 1 = 6/7 concurrent, bit 2 = the critics of 5 / 7 on four lanes, bit 3 = filter / bias gradients on the lanes' leaf streams
    c.lanes_mask = atoi(e);
    c.lanes_on = c.lanes_mask != 0;
  }
  c.leaf_on = (c.lanes_mask & 8) != 0;
  if (const char* e = getenv("EG_DEFER_REDUCE")) c.defer_on = atoi(e) != 0;
  Ctx* cp = &c;
  c.tape.after = [cp]() { flush_deferred(*cp); };
}

// The lanes beyond the four train_step schedules on (inference slots 4 .. 7): a main stream, scratch and the join event, created
// the first time a slot needs them -- after every stream of build_context, whose creation order the queue placement relies on
void ensure_lane(Ctx& c, int lane) {
  EG_CHECK(lane >= 0 && lane < Ctx::kLanes, "lane %d out of range", lane);
  Ctx::Lane& l = c.lanes[lane];
  if (l.stream) return;
  EG_HIP(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
  EG_HIP(hipMalloc(&l.scratch, c.scratch_floats * 4));
  EG_HIP(hipEventCreateWithFlags(&l.done, hipEventDisableTiming));
}

void use_lane(Ctx& c, int lane) {
  if (lane == c.cur_lane) return;
  // (checked BEFORE anything is swapped: a throw from here must leave the context exactly as it was)
  EG_CHECK(!c.leaf_dirty, "use_lane with un-joined leaf work on lane %d", c.cur_lane);
  c.arena.swap(c.lanes[c.cur_lane].arena);   // park the active arena in its slot
  c.arena.swap(c.lanes[lane].arena);         // bring lane's arena in (its slot keeps the empty placeholder)
  c.stream = c.lanes[lane].stream;
  c.scratch = c.lanes[lane].scratch;
  c.cur_lane = lane;
}

// Every C-ABI entry point that enqueues work starts here (capi.cpp: EG_API_BEGIN_CTX).  An exception in the middle of a
// backward sweep (a HIP error, "variable received no gradient", ...) unwinds past leaf_join, i.e. it can leave leaf_dirty
// set, the context's stream pointing at a lane's LEAF stream (LeafScope restores it, an error inside run_deferred does
// not) and another lane active.  The next call must not inherit that: drain the device (whatever the failed call
// enqueued is finished or failed), put the active lane back on its main stream, drop the queued reductions of the
// abandoned sweep, then return to lane 0.
void api_enter(Ctx& c) {
  const bool from_infer_run = c.inf_entering;
  c.inf_entering = false;
  if (c.inf_inflight && !from_infer_run) {   // inference calls in flight on other lanes (capi.cpp: eg_infer_run): drain them
    (void)hipDeviceSynchronize();
    c.inf_inflight = false;
  }
  if (c.leaf_dirty || c.stream != c.lanes[c.cur_lane].stream) {
    (void)hipDeviceSynchronize();
    (void)hipGetLastError();
    c.stream = c.lanes[c.cur_lane].stream;
    c.deferred.clear();
    c.leaf_dirty = false;
  }
  use_lane(c, 0);
  c.infer_side = -1;
  c.feeds_in = FEED_NONE;
  c.prelu_plane_only = false;
  c.conv_out_into = nullptr;
  c.pool_epi = Ctx::PoolEpi();
}

// ---------------------------------------------------------------------------------------------
// networks
// ---------------------------------------------------------------------------------------------
// generator.py:35-74 (_convnet): linear -> reshape -> BatchNorm(!) -> ReLU -> 3x[deconv,IN,ReLU] -> deconv -> tanh
// the block norm with E_norm / G_norm / D_norm = 'batch' (normalization.py:19-25): BatchNorm over this evaluation's rows + the
// block's activation.  The moving statistics advance `times` times; they never enter a result (is_training=True, line 24).
static T block_bn(Ctx& c, T x, const std::string& scope, int a, int times, T* stats_out = nullptr) {
  const std::string b = scope + "/batch_norm/BatchNorm/";
  return bn_act(c, x, P(c, b + "gamma"), P(c, b + "beta"), P(c, b + "moving_mean"), P(c, b + "moving_variance"), a, times, stats_out);
}

T generator_forward(Ctx& c, int group, T zin, int bn_times, T* bn_stats_out, float* out_into) {
  const Config& cfg = c.cfg;
  const std::string p = kGroupPrefix[group];
  const int gf = cfg.gf, h16 = cfg.H / 16, w16 = cfg.W / 2 / 16;
  const int feat = gf * 8 * h16 * w16;
  T z_ = linear(c, zin, wref_param(c, P(c, p + "/g_lin_0/Matrix"), 1, zin->c, feat), P(c, p + "/g_lin_0/bias"));
  T x = reshape(c, z_, zin->n, h16, w16, gf * 8);
  // (a one-launch form of linear + BatchNorm + ReLU for forward-only graphs was built and measured in round 6: 43 us against
  //  18 + 12 for the two launches it replaced -- 64 blocks of a latency-bound gather of 32-byte filter pieces -- and its batch
  //  statistics, the same numbers in another summation order, broke the bit-equality of sub-step 6's forward with sub-step 7's:
  //  removed, docs/experiments.md)
  x = bn_relu(c, x, P(c, p + "/batch_norm/BatchNorm/gamma"), P(c, p + "/batch_norm/BatchNorm/beta"),
              P(c, p + "/batch_norm/BatchNorm/moving_mean"), P(c, p + "/batch_norm/BatchNorm/moving_variance"),
              bn_times, bn_stats_out);
  const int ch[5] = {gf * 8, gf * 4, gf * 2, gf, cfg.c_dim};
  for (int i = 1; i <= 4; ++i) {
    const std::string b = p + "/g_dconv_" + std::to_string(i) + "/deconv2d/";
    c.feeds_in = i < 4 && !cfg.g_bn && !cfg.g_nonorm;   // deconv_block: the InstanceNorm is the transposed convolution's only reader
    x = deconv2d(c, x, wref_param(c, P(c, b + "w"), 25, ch[i], ch[i - 1]), P(c, b + "b"), ch[i], i == 4 ? ACT_TANH : 0,
                 i == 4 ? out_into : nullptr);
    // (bf16 mode, forward-only graph: the next transposed convolution stages the bf16 plane -- the fp32 copy is not written;
    //  g_dconv_4 gathers fp32, so layer 3's output keeps both)
    if (i < 4 && cfg.g_bn) {   // G_norm = 'batch' (generator.py:54-68 through conv.py:124-130)
      T bst = nullptr;
      x = block_bn(c, x, p + "/g_dconv_" + std::to_string(i), ACT_RELU, bn_times, &bst);
      c.gen_blk_stats[i - 1] = bst;
    } else if (i < 4 && cfg.g_nonorm) {   // G_norm = None (normalization.py:26-27): deconv -> ReLU
      x = act(c, x, ACT_RELU);
    } else if (i < 4)
      x = in_act(c, x, ACT_RELU, nullptr, /*plane_only=*/cfg.bf16 && !x->needs_grad && i < 3);
  }
  if (x->fused_act == ACT_TANH) {   // (forward-only graph: the last layer's store applied it)
    x->fused_act = 0;
    return x;
  }
  T out = act(c, x, ACT_TANH);
  if (out_into && out->d != out_into) launch_copy(out->d, out_into, out->numel(), c.stream);
  return out;
}

// discriminator.py:58-81 (_convnet).  With tangent_of != null this evaluates the forward-mode
// tangent of the same network at the saved primal activations (convs are linear, so the tangent
// of a conv is the conv; IN / LeakyReLU are linearised).
T disc_forward(Ctx& c, int group, T x, DiscSave* save, const DiscSave* tangent_of) {
  const Config& cfg = c.cfg;
  const std::string p = kGroupPrefix[group];
  const int df = cfg.df;
  const int ch[5] = {cfg.c_dim, df, df * 2, df * 4, df * 8};
  for (int i = 0; i < 4; ++i) {
    WRef* w = wref_param(c, P(c, p + "/" + kDiscNames[i] + "/conv2d/w"), 16, ch[i], ch[i + 1]);
    // (first block: conv + LeakyReLU, fused where the small-C kernel runs the conv; `pre` is then the activated output,
    //  which is all the tangent pass and the backward need of it)
    c.feeds_in = i > 0 && !tangent_of && !cfg.d_bn && !cfg.d_nonorm;   // conv_block: the InstanceNorm is the convolution's only forward reader
    c.conv_plane_only_next = cfg.bf16 && i == 0 && !tangent_of && !save;   // (no tangent pass will read this output's values)
    T pre = conv2d(c, x, w, nullptr, 4, 2, PAD_SAME, nullptr, (i == 0 && !tangent_of) ? ACT_LRELU : 0);
    T out, st = nullptr;
    if (cfg.d_bn && i > 0) {   // D_norm = 'batch' (discriminator.py:63-72 through conv.py:61-67)
      const std::string scope = p + "/" + kDiscNames[i];
      out = !tangent_of ? block_bn(c, pre, scope, ACT_LRELU, c.bn_times_block, &st)
                        : bn_act_tangent(c, pre, tangent_of->pre[i], tangent_of->out[i], tangent_of->stats[i],
                                         P(c, scope + "/batch_norm/BatchNorm/gamma"), ACT_LRELU);
      if (save && !tangent_of) {
        save->pre[i] = pre;
        save->out[i] = out;
        save->stats[i] = st;
      }
    } else if (cfg.d_nonorm && i > 0) {   // D_norm = None (normalization.py:26-27): conv -> LeakyReLU, as block 1
      out = !tangent_of ? act(c, pre, ACT_LRELU) : act_tangent(c, pre, tangent_of->out[i], ACT_LRELU);
      if (save && !tangent_of) {
        save->pre[i] = pre;
        save->out[i] = out;
      }
    } else if (!tangent_of) {
      out = (i == 0) ? pre : in_act(c, pre, ACT_LRELU, &st, /*plane_only=*/cfg.bf16 && i < 3);   // (layer 3 feeds the linear head)
      if (save) {
        save->pre[i] = pre;
        save->out[i] = out;
        save->stats[i] = st;
      }
    } else {
      out = (i == 0) ? act_tangent(c, pre, tangent_of->out[0], ACT_LRELU)
                     : in_act_tangent(c, pre, tangent_of->pre[i], tangent_of->out[i], tangent_of->stats[i], ACT_LRELU);
    }
    x = out;
  }
  return linear1(c, x, P(c, p + "/d_linear_5/Matrix"), tangent_of ? nullptr : P(c, p + "/d_linear_5/bias"));
}

// encoder.py:54-84 (_resnet) with conv.py:70-85 (residual, REFLECT pad, bias=True)
T encoder_forward(Ctx& c, T x, const float* eps_dev, T* mu_out, T* ls_out, T* z1_out) {
  const Config& cfg = c.cfg;
  T E = conv2d(c, x, wref_param(c, P(c, "E/e_resnet_64_0/conv2d/w"), 16, cfg.c_dim, 64),
               P(c, "E/e_resnet_64_0/conv2d/b"), 4, 2, PAD_SAME, nullptr, ACT_RELU);
  int cin = 64, i = 1;
  for (int n : encoder_filters(cfg)) {
    const std::string b = "E/e_resnet_" + std::to_string(n) + "_" + std::to_string(i++);
    const int main_lane = c.cur_lane;
    if (c.infer_side >= 0) EG_HIP(hipEventRecord(c.ev_infer[main_lane], c.stream));   // the block's input is complete here
    c.feeds_in = !cfg.e_bn && !cfg.e_nonorm;   // residual: both 3x3 convolutions are read by their InstanceNorm alone
    T o1 = conv2d(c, E, wref_param(c, P(c, b + "/res1/conv2d/w"), 9, cin, n), P(c, b + "/res1/conv2d/b"), 3, 1,
                  PAD_REFLECT);
    // E_norm = 'batch' (encoder.py:67 through conv.py:70-85): BatchNorm in both halves of the block
    // E_norm = None (normalization.py:26-27): conv -> ReLU -> conv
    o1 = cfg.e_bn ? block_bn(c, o1, b + "/res1", ACT_RELU, c.bn_times_block)
         : cfg.e_nonorm ? act(c, o1, ACT_RELU)
                  : in_act(c, o1, ACT_RELU, nullptr, /*plane_only=*/cfg.bf16 && !o1->needs_grad);   // (forward-only graph: res2 stages the plane)
    c.feeds_in = !cfg.e_bn && !cfg.e_nonorm;
    T o2 = conv2d(c, o1, wref_param(c, P(c, b + "/res2/conv2d/w"), 9, n, n), P(c, b + "/res2/conv2d/b"), 3, 1,
                  PAD_REFLECT);
    if (!cfg.e_nonorm) o2 = cfg.e_bn ? block_bn(c, o2, b + "/res2", ACT_NONE, c.bn_times_block) : in_act(c, o2, ACT_NONE);
    if (c.infer_side >= 0) {
      use_lane(c, c.infer_side);
      EG_HIP(hipStreamWaitEvent(c.stream, c.ev_infer[main_lane], 0));
    }
    T sc = conv2d(c, E, wref_param(c, P(c, b + "/shortcut/conv2d/w"), 1, cin, n), P(c, b + "/shortcut/conv2d/b"),
                  1, 1, PAD_REFLECT);
    if (c.infer_side >= 0) {
      EG_HIP(hipEventRecord(c.lanes[c.infer_side].done, c.stream));
      use_lane(c, main_lane);
      EG_HIP(hipStreamWaitEvent(c.stream, c.lanes[c.infer_side].done, 0));
    }
    E = add_relu_pool2(c, sc, o2);
    cin = n;
  }
  EG_CHECK(E->h <= 8 && E->w <= 8, "encoder tail expects <= 8x8 before the 8x8 SAME average pool");
  if (z1_out) *z1_out = nullptr;
  {
    // forward-only graph (inference, the post-update evaluation): relu + mean, both FC8 linears and the reparametrisation of the
    // call's one or two eps draws in one launch (k_elem.hip: enc_head_kernel)
    static const bool head_on = dbg_env("EG_NO_ENC_HEAD") == nullptr;
    Param *wm = P(c, "E/FC8_mu/w"), *bm = P(c, "E/FC8_mu/b"), *ws = P(c, "E/FC8_sigma/w"), *bs = P(c, "E/FC8_sigma/b");
    if (head_on && cfg.z_dim % 4 == 0 && !E->needs_grad && !wm->want_grad && !ws->want_grad && !bm->want_grad && !bs->want_grad) {
      const int Z = cfg.z_dim;
      T mu = new_tensor(c, E->n, 1, 1, Z, false), ls = new_tensor(c, E->n, 1, 1, Z, false), z0 = new_tensor(c, E->n, 1, 1, Z, false);
      T z1 = z1_out ? new_tensor(c, E->n, 1, 1, Z, false) : nullptr;
      // (the linear's bf16 rule, ops.cpp attach_bf16 / oracle.bf16_conv_rule: contracted channels % 32 == 0, > 16 produced, % 4 == 0)
      const int round = (cfg.bf16 && cin % 32 == 0 && Z > 16 && Z % 4 == 0) ? 1 : 0;
      launch_enc_head(E->d, wm->w, bm->w, ws->w, bs->w, eps_dev, mu->d, ls->d, z0->d, z1 ? z1->d : nullptr, E->n, E->h * E->w, cin, Z, round,
                      c.stream);
      if (mu_out) *mu_out = mu;
      if (ls_out) *ls_out = ls;
      if (z1_out) *z1_out = z1;
      return z0;
    }
  }
  T feat = gmean(c, E, true);  // relu + avg_pool(8x8, SAME) == mean of the valid elements
  T mu = linear(c, feat, wref_param(c, P(c, "E/FC8_mu/w"), 1, cin, cfg.z_dim), P(c, "E/FC8_mu/b"));
  T ls = linear(c, feat, wref_param(c, P(c, "E/FC8_sigma/w"), 1, cin, cfg.z_dim), P(c, "E/FC8_sigma/b"));
  if (mu_out) *mu_out = mu;
  if (ls_out) *ls_out = ls;
  return reparam(c, mu, ls, eps_dev);
}

// classifier.py:12-119 + conv.py:133-357.  The reference runs NCHW; channels-last is the same
// arithmetic (all spatial reductions are per (n,c)), so the engine stays NHWC end to end.
T classifier_forward(Ctx& c, T x) {
  if (!c.sn_prepared) {
    sn_prepare(c);
    c.sn_prepared = true;
  }
  // input pyramid by 2x2 mean pooling (classifier.py:27-35); levels 64,32,16,8 are used
  if (x->needs_grad) {
    // (pushed first = runs LAST in this forward's backward range: the image pyramid's gradient chain lives on the leaf
    //  stream, see below; whatever follows on the main stream -- the resize / concat backward, the generator's -- reads and
    //  accumulates into d(x) only behind it)
    Ctx* cp = &c;
    c.tape.push([cp]() { leaf_join(*cp); });
  }
  T level[4];
  level[0] = x;
  for (int i = 1; i < 4; ++i) level[i] = pool2(c, level[i - 1]);
  // Gradients towards the 3-channel pyramid (sub-steps 5 / 7 only: d(loss)/d(G2(z))) are leaves of the hidden-state chain:
  // every producer -- the image convolutions' and gate image rows' input gradients, the pyramid's pool backward -- runs
  // on the lane's leaf stream, in tape order, and the chain  dgrad -> PReLU / gate backward -> dgrad  does not wait for
  // these thin, HBM-bound launches
  if (x->needs_grad)
    for (int i = 0; i < 4; ++i) level[i]->leaf_grad = true;
  // (conv + PReLU pairs go through conv2d's post_prelu: the PReLU is a second store of the convolution's epilogue where its
  //  kernel has one; the PReLU of a unit's OUTPUT -- the next unit's prelu(h), the final one -- is a second store of pool2_add)
  T h = conv2d(c, x, wref_sn(c, P(c, "D2/Conv/weights")), P(c, "D2/Conv/biases"), 7, 1, PAD_SAME, nullptr, 0,
               P(c, "D2/Conv/prelu/param"));
  T hp = nullptr;
  int hidden = 8;
  for (int u = 1; u <= 4; ++u) {
    const std::string b = "D2/mru_conv_unit_t_" + std::to_string(u) + "_layer_0";
    const int depth = kMruDepth[u - 1];
    T xin = level[u - 1];
    if (!hp) hp = prelu(c, h, P(c, b + "/norm_activation_in/prelu/param"), 0.f);
    // update gate on concat([prelu(h), x]) (conv.py:185-198) without materialising the concat:
    // the SN filter is split by input-channel rows into a hidden part and an image part.
    Param* wg = P(c, b + "/update_gate/weights");
    WRef* full = wref_sn(c, wg);
    const int ct = hidden + 3;
    SNWeight& sw = c.sn[c.sn_index.at(wg)];   // its two parts are refreshed together with W / sigma (ops.cpp: sn_prepare)
    EG_CHECK(sw.split_rows == hidden, "update gate %s: split rows %d vs hidden %d", wg->name.c_str(), sw.split_rows, hidden);
    float *w_h = sw.part[0], *w_x = sw.part[1];
    float *g_h = nullptr, *g_x = nullptr;
    bool *f_h = nullptr, *f_x = nullptr;
    c.flags.push_back(false);
    f_h = &c.flags.back();
    c.flags.push_back(false);
    f_x = &c.flags.back();
    if (full->want_grad) {
      g_h = c.arena.allocf((size_t)9 * hidden * hidden);
      g_x = c.arena.allocf((size_t)9 * 3 * hidden);
      Ctx* cp = &c;
      c.tape.push([=]() {  // runs after both partial wgrads (pushed before them)
        Ctx& c = *cp;
        if (!c.param_grad_enabled || !*f_h || !*f_x) return;
        flush_deferred(c);   // the two partial filter gradients are reduced before their columns are copied
        launch_copy_cols(g_h, full->g, 9, 1, hidden, ct, 0, hidden, 1, 0, c.stream);
        launch_copy_cols(g_x, full->g, 9, 1, 3, ct, hidden, hidden, 1, 0, c.stream);
        *full->g_set = true;
      });
    }
    WRef* wr_h = wref_raw(c, w_h, g_h, f_h, full->want_grad, 9, hidden, hidden, wg->name + "[hidden rows]");
    WRef* wr_x = wref_raw(c, w_x, g_x, f_x, full->want_grad, 9, 3, hidden, wg->name + "[image rows]");
    int pi = 0;
    for (WRef* wr : {wr_h, wr_x}) {   // persistent per-tap transposes / limb planes, like Wbar's own
      wr->extT = sw.partT[pi];
      wr->extT_valid = &sw.partT_valid[pi];
      wr->extL = sw.partL[pi];
      ++pi;
    }
    // image rows (+ bias) first, the hidden rows add to them: a K-split launch of the hidden-state convolution then leaves
    // its slabs to the min-max pass (FEED_MINMAX), which sums them on top of the image rows' output
    T rg = conv2d(c, xin, wr_x, P(c, b + "/update_gate/biases"), 3, 1, PAD_SAME);
    c.feeds_in = FEED_MINMAX;
    conv2d(c, hp, wr_h, nullptr, 3, 1, PAD_SAME, rg);
    rg = minmax(c, rg, 0.2f);   // lrelu (activation.py:30-32) + the min-max gate of conv.py:197-198, one pass each way
    T img = conv2d(c, xin, wref_sn(c, P(c, b + "/Conv/weights")), P(c, b + "/Conv/biases"), 3, 1, PAD_SAME);
    // (bf16 mode: the gate's output is read by Conv_1's forward and filter gradient only -- plane-fed, so only its plane is stored)
    T hin = gate_prelu(c, h, rg, img, P(c, b + "/norm_activation_merge_1/prelu/param"), c.cfg.bf16);   // conv.py:207-209
    c.prelu_plane_only = c.cfg.bf16;   // (bf16 mode: prelu(Conv_1) is read by Conv_2's forward and filter gradient only -- plane-fed)
    T c1 = conv2d(c, hin, wref_sn(c, P(c, b + "/Conv_1/weights")), P(c, b + "/Conv_1/biases"), 3, 1, PAD_SAME, nullptr, 0,
                  P(c, b + "/Conv_1/prelu/param"));   // conv.py:217-220
    static const bool proj_full = dbg_env("EG_PROJ_FULL_RES") != nullptr;   // A/B: the reference's evaluation order
    T c2 = proj_full ? conv2d(c, c1, wref_sn(c, P(c, b + "/Conv_2/weights")), P(c, b + "/Conv_2/biases"), 3, 1, PAD_SAME) : nullptr;
    // conv.py:229-243: ht_new = mean_pool(conv1x1(ht) + h_new).  The projection is a per-pixel linear map (+ bias) and the
    // pool a per-channel mean over 2x2 pixels, so mean_pool(conv1x1(ht)) == conv1x1(mean_pool(ht)): the 1x1 convolution
    // (forward, input gradient, filter gradient) runs on a quarter of the pixels and its full-resolution output -- as
    // large as h_new -- is never written or read.  Same value up to fp32 summation order.
    hp = nullptr;
    if (proj_full) {
      T pf = conv2d(c, h, wref_sn(c, P(c, b + "/Conv_3/weights")), P(c, b + "/Conv_3/biases"), 1, 1, PAD_SAME);
      h = add_pool2(c, pf, c2);   // pf and c2 are conv outputs with no other consumer
    } else {
      // (measured and not kept: this projection and the image convolution above on the leaf stream beside the gate chain
      //  -- 1,530 -> 1,476 images/s, profiles/r03f_ab_forward_leaves_negative.txt: thin persistent kernels beside the
      //  one-block-per-CU matrix kernels of the same chain take CUs from them)
      // (the projection comes FIRST: Conv_2's epilogue pools its own output and adds it, so that h_new at full resolution -- as
      //  large as Conv_1's output -- is never stored where the kernel can do that; a K-split launch leaves slabs to pool2_add)
      T proj = conv2d(c, pool2(c, h), wref_sn(c, P(c, b + "/Conv_3/weights")), P(c, b + "/Conv_3/biases"), 1, 1, PAD_SAME);
      // the PReLU of the unit's output rides along: the next unit's prelu(h) (conv.py:185), the final one after unit 4 (:353-355)
      Param* next_leak = u < 4 ? P(c, "D2/mru_conv_unit_t_" + std::to_string(u + 1) + "_layer_0/norm_activation_in/prelu/param")
                               : P(c, "D2/mru_conv_unit_last_norm/prelu/param");
      // (bf16 mode: the next unit's prelu(h) is read by the gate's hidden-state convolution only; the LAST one feeds the global mean)
      h = conv2d_pool2_add(c, c1, wref_sn(c, P(c, b + "/Conv_2/weights")), P(c, b + "/Conv_2/biases"), 3, proj, next_leak, &hp,
                           c.cfg.bf16 && u < 4);
    }
    hidden = depth;
  }
  h = hp ? hp : prelu(c, h, P(c, "D2/mru_conv_unit_last_norm/prelu/param"), 0.f);  // conv.py:353-355
  T feat = gmean(c, h, false);
  return linear(c, feat, wref_sn(c, P(c, "D2/fully_connected/weights")), P(c, "D2/fully_connected/biases"));
}

// ---------------------------------------------------------------------------------------------
// sub-steps
// ---------------------------------------------------------------------------------------------
static T make_zin(Ctx& c) {
  const Config& cfg = c.cfg;
  if (!cfg.multiclasses()) return wrap_tensor(c, c.d_z, cfg.B, 1, 1, cfg.z_dim);
  T zin = new_tensor(c, cfg.B, 1, 1, cfg.g_in(), false);
  launch_z_onehot(c.d_z, zin->d, cfg.B, cfg.z_dim, cfg.num_classes, c.stream);
  return zin;
}

// Data parallel: sum the group's flat gradient over the ranks (RCCL) and average inside RMSProp.  Both run on
// the side stream behind an event, so the exchange of sub-step k overlaps the compute of sub-step k+1 whenever
// that one does not read this group's weights (sub-steps 1-4 touch disjoint groups and G is frozen until 5;
// E's update overlaps sub-step 7) -- run_substep() waits for exactly the groups it reads.
// The stream a group's averaging RMSProp runs on behind its all-reduce: the communication stream itself -- or, when that stream
// carries a CU mask (EG_COMM_CUS), an unmasked side stream ordered behind the collective by an event: the optimizer is an
// HBM-bound pass over up to 17 M parameters that must not be confined to the exchange's few compute units
static hipStream_t comm_apply_stream(Ctx& c, int group) {
  if (!c.comm_cus) return c.comm_stream;
  if (!c.apply_stream) EG_HIP(hipStreamCreateWithFlags(&c.apply_stream, hipStreamNonBlocking));
  EG_HIP(hipEventRecord(c.ev_ready[group], c.comm_stream));     // (ev_ready[group] has done its first job: the exchange is enqueued behind it)
  EG_HIP(hipStreamWaitEvent(c.apply_stream, c.ev_ready[group], 0));
  return c.apply_stream;
}

static void apply_group(Ctx& c, int group) {
  NetGroup& g = c.groups[group];
  if (group == G_G1) c.gen_cache_valid[0] = false;
  if (group == G_G2) c.gen_cache_valid[1] = false;
  for (Param* p : g.params) p->invalidate_derived();
  for (Param* p : g.params)
    if (p->trainable && !p->dead) EG_CHECK(p->g_set, "variable %s received no gradient", p->name.c_str());
  if (c.nccl_comm) {
    EG_HIP(hipEventRecord(c.ev_ready[group], c.stream));
    EG_HIP(hipStreamWaitEvent(c.comm_stream, c.ev_ready[group], 0));
    comm_allreduce(c, g.g, g.n_live, c.comm_stream, group);
    hipStream_t as = comm_apply_stream(c, group);   // (the optimizer does not run under the exchange's CU mask)
    launch_rmsprop(g.w, g.g, g.ms, g.mom, g.n_live, c.cfg.lr, 1.f / c.nranks, as);
    EG_HIP(hipEventRecord(c.ev_applied[group], as));
    c.pending[group] = true;
    return;
  }
  launch_rmsprop(g.w, g.g, g.ms, g.mom, g.n_live, c.cfg.lr, 1.f, c.stream);
}

// Sub-steps 5 / 7 optimise G1 and G2 together (quirk N9): their gradient buffers are adjacent, so under data parallelism
// both travel in ONE all-reduce (2 x 5.25 M floats); RMSProp stays one launch per group (the weight buffers also hold
// the BatchNorm moving statistics between the two live ranges).
static void apply_generators(Ctx& c) {
  if (!c.nccl_comm) {
    apply_group(c, G_G1);
    apply_group(c, G_G2);
    return;
  }
  NetGroup &g1 = c.groups[G_G1], &g2 = c.groups[G_G2];
  c.gen_cache_valid[0] = c.gen_cache_valid[1] = false;
  for (NetGroup* g : {&g1, &g2})
    for (Param* p : g->params) {
      p->invalidate_derived();
      if (p->trainable && !p->dead) EG_CHECK(p->g_set, "variable %s received no gradient", p->name.c_str());
    }
  EG_CHECK(g2.g == g1.g + g1.n_train, "generator gradient buffers are not adjacent");
  EG_HIP(hipEventRecord(c.ev_ready[G_G1], c.stream));
  EG_HIP(hipStreamWaitEvent(c.comm_stream, c.ev_ready[G_G1], 0));
  comm_allreduce(c, g1.g, g1.n_train + g2.n_live, c.comm_stream, G_G1);
  hipStream_t as = comm_apply_stream(c, G_G1);
  launch_rmsprop(g1.w, g1.g, g1.ms, g1.mom, g1.n_live, c.cfg.lr, 1.f / c.nranks, as);
  launch_rmsprop(g2.w, g2.g, g2.ms, g2.mom, g2.n_live, c.cfg.lr, 1.f / c.nranks, as);
  EG_HIP(hipEventRecord(c.ev_applied[G_G1], as));
  EG_HIP(hipEventRecord(c.ev_applied[G_G2], as));
  c.pending[G_G1] = c.pending[G_G2] = true;
}

// WGAN-GP discriminator step (edgegan.py:278-309, functional.py:26-33, edgegan.py:32-42).
// The penalty's second-order term is evaluated as reverse-over-forward: with g = d/dx sum_b[sigmoid(D)+D]
// (quirk N10) and v = dL_gp/dg held constant, dL_gp/dtheta = d/dtheta <g, v> = d/dtheta JVP_v[sigmoid(D)+D],
// so a tangent pass of D along v at x_hat followed by ONE reverse sweep over (primal, tangent) gives
// the exact gradient with first-order kernels only.
static void disc_substep(Ctx& c, int group, T real, T fake, const float* alpha_dev, int slot_d, int slot_gp,
                         bool apply) {
  const Config& cfg = c.cfg;
  const int B = cfg.B;
  set_want_grad(c, group, true);
  T xhat = new_tensor(c, real->n, real->h, real->w, real->c, true);
  const size_t per = xhat->numel() / B;
  launch_blend(real->d, fake->d, alpha_dev, xhat->d, B, per, c.stream);
  // --- primal pass on x_hat and first-order gradient g ---
  DiscSave sv;
  T d_hat = disc_forward(c, group, xhat, &sv, nullptr);
  const size_t mark = c.tape.ops.size();
  seed_grad(c, d_hat);
  launch_gp_head(d_hat->d, nullptr, d_hat->g, nullptr, B, 0, c.stream);
  c.param_grad_enabled = false;
  c.tape.backward_range(0, mark);
  c.param_grad_enabled = true;
  const float* g = xhat->g;
  reset_grads(c);
  xhat->needs_grad = false;  // the final sweep does not need d/dx_hat again
  float* v = c.arena.allocf(xhat->numel());
  float* norms = c.arena.allocf(B);
  launch_gp_v(g, v, norms, c.d_losses + slot_gp, B, per, cfg.lambda_gp, c.stream);
  // --- real / fake pass (stacked 2B; D_norm = 'batch': two evaluations of B rows -- the reference calls the discriminator on
  //     each input separately, edgegan.py:236-262, and BatchNorm's statistics are per call) ---
  T d_rf = nullptr, d_real = nullptr, d_fake = nullptr;
  if (cfg.d_bn) {
    d_real = disc_forward(c, group, real, nullptr, nullptr);
    d_fake = disc_forward(c, group, fake, nullptr, nullptr);
    launch_mean_diff(d_fake->d, d_real->d, c.d_losses + slot_d, B, c.stream);
  } else {
    T rf = new_tensor(c, 2 * B, real->h, real->w, real->c, false);
    launch_copy(real->d, rf->d, real->numel(), c.stream);
    launch_copy(fake->d, rf->d + real->numel(), fake->numel(), c.stream);
    d_rf = disc_forward(c, group, rf, nullptr, nullptr);
    launch_mean_diff(d_rf->d + B, d_rf->d, c.d_losses + slot_d, B, c.stream);
  }
  // --- tangent pass along v ---
  T tv = wrap_tensor(c, v, B, real->h, real->w, real->c);
  T td = disc_forward(c, group, tv, nullptr, &sv);
  // --- seeds and the single reverse sweep ---
  if (cfg.d_bn) {
    seed_grad(c, d_real);
    launch_fill(d_real->g, -1.f / B, B, c.stream);
    seed_grad(c, d_fake);
    launch_fill(d_fake->g, 1.f / B, B, c.stream);
  } else {
    seed_grad(c, d_rf);
    launch_fill(d_rf->g, -1.f / B, B, c.stream);
    launch_fill(d_rf->g + B, 1.f / B, B, c.stream);
  }
  seed_grad(c, d_hat);
  seed_grad(c, td);
  launch_gp_head(d_hat->d, td->d, d_hat->g, td->g, B, 1, c.stream);
  c.tape.backward();
  if (apply) apply_group(c, group);
  set_want_grad(c, group, false);
}

// The same sub-step with the discriminator's four passes STACKED along the batch: rows [0,B) x_hat (primal), [B,2B) real,
// [2B,3B) fake, [3B,4B) the tangent along v.  The first-order pass needs x_hat's forward before v exists, so the forward runs
// in two phases (B rows, then 3B rows in one launch per layer); the reverse sweep then runs ONE filter-gradient and ONE
// input-gradient launch per layer over all 4B rows -- convolutions are linear, the tangent of a convolution is the
// convolution -- where the tape's sweep ran three of each on B / 2B / B rows and accumulated.  Only the InstanceNorm +
// LeakyReLU stages see the row classes: in_fwd / in_bwd on the primal and real|fake rows, the linearisation (in_tangent_fwd)
// and its reverse on the tangent rows.  The linear tape cannot express this order (per layer: every row class's norm backward,
// then one convolution backward), so the sweep is written out here.  Same kernels, same mathematics as disc_substep; sums over
// the batch are taken in one launch instead of three (fp32 summation order differs).
// Round 6: the branch at the reference's cost (6F - F0 instead of 7F - 2F0, functional.py:26-29 / edgegan.py:38-42).  The
// tangent t = J v enters the loss through <s(d), t_out>, s = 1 + sigmoid'(d), so the cotangent of the tangent at layer i is
// J_(i->out)^T s(d) -- exactly the FIRST-ORDER gradient delta_i that phase 1 computes with the same seed (gp_head mode 0 and the
// tangent seed of mode 1 are the same expression).  Phase 1 therefore stores its delta(out_i) in the TANGENT rows of the
// stacked gradient tensors, and the reverse sweep's input-gradient launches run over the 3B evaluated rows only: one
// discriminator input-gradient chain (layers 3, 2, 1 over B rows) less per sub-step.  Measured per layer (f32x3, forward +
// both gradients): d_conv_4 of the patch discriminators 1.08 -> 0.96 ms, of the joint one 0.64 -> 0.51
// (tools/gpu_stack_probe.py, profiles/r05g_*).
static void disc_substep_stacked(Ctx& c, int group, T real, T fake, const float* alpha_dev, int slot_d, int slot_gp,
                                 bool apply) {
  const Config& cfg = c.cfg;
  const int B = cfg.B, R = 4 * B;
  const std::string p = kGroupPrefix[group];
  set_want_grad(c, group, true);
  const int H = real->h, W = real->w, C0 = real->c;
  const int ch[5] = {C0, cfg.df, cfg.df * 2, cfg.df * 4, cfg.df * 8};
  WRef* w[4];
  for (int i = 0; i < 4; ++i) w[i] = wref_param(c, P(c, p + "/" + kDiscNames[i] + "/conv2d/w"), 16, ch[i], ch[i + 1]);
  Param* lw = P(c, p + "/d_linear_5/Matrix");
  Param* lb = P(c, p + "/d_linear_5/bias");
  // stacked activations: X = the four inputs, pre[i] / out[i] = layer i before / after its norm + activation
  T X = new_tensor(c, R, H, W, C0, false);
  X->g = c.arena.allocf((size_t)B * H * W * C0);   // (only the x_hat rows [0,B) ever receive a gradient: view_rows(X, 0, B))
  T pre[4], out[4], st[4];
  unsigned short* gpl[4];
  int hh = H, ww = W;
  const bool planes = cfg.bf16;
  for (int i = 0; i < 4; ++i) {
    hh = (hh + 1) / 2;
    ww = (ww + 1) / 2;
    pre[i] = new_tensor(c, R, hh, ww, ch[i + 1], true);
    out[i] = i == 0 ? pre[0] : new_tensor(c, R, hh, ww, ch[i + 1], true);     // layer 0: the activation is applied in place
    st[i] = i == 0 ? nullptr : new_tensor(c, 3 * B, 1, 1, 2 * ch[i + 1], false);
    if (planes && ch[i + 1] % 32 == 0 && out[i]->numel() * 2 < (1ull << 31)) {   // the next convolution's operand plane (bf16 mode)
      out[i]->d_bf = reinterpret_cast<unsigned short*>(c.arena.allocf((out[i]->numel() + 1) / 2));
      out[i]->d_bf_stream = nullptr;
      // layers 1 and 2 are read by convolutions only (their own backward recomputes act'(out) from pre): the plane alone
      static const bool po_on = dbg_env("EG_NO_PLANE_ONLY_IN") == nullptr;
      out[i]->d_plane_only = po_on && (i == 1 || i == 2);
    }
    pre[i]->g = c.arena.allocf(pre[i]->numel());
    if (i > 0) out[i]->g = c.arena.allocf(out[i]->numel());
    // bf16 mode: the gradient planes both gradient convolutions of layer i read, stored by the norm backward kernels
    gpl[i] = nullptr;
    if (planes && i > 0 && ch[i + 1] % 32 == 0 && pre[i]->numel() * 2 < (1ull << 31))
      gpl[i] = reinterpret_cast<unsigned short*>(c.arena.allocf((pre[i]->numel() + 1) / 2));
  }
  const size_t per = (size_t)H * W * C0;
  auto rows = [&](T t, int n0, int n) { return view_rows(c, t, n0, n); };
  launch_blend(real->d, fake->d, alpha_dev, X->d, B, per, c.stream);
  launch_copy(real->d, X->d + (size_t)B * per, (size_t)B * per, c.stream);
  launch_copy(fake->d, X->d + (size_t)2 * B * per, (size_t)B * per, c.stream);
  const int K = (int)(out[3]->numel() / R);
  EG_CHECK((size_t)K == lw->numel, "d_linear_5 K mismatch %d vs %zu", K, lw->numel);
  float* dval = c.arena.allocf(R);     // the head's outputs: d(x_hat) | d(real) | d(fake) | tangent
  float* dgrad = c.arena.allocf(R);    // ... and their gradients
  // one forward pass over rows [n0, n0 + n): rows below 3B are evaluated points, rows from 3B on the tangent along v
  auto forward = [&](int n0, int n) {
    const int ne = std::min(n0 + n, 3 * B) - n0;       // evaluated rows in this pass
    const int nt = n - ne;                              // tangent rows in this pass (0 or B)
    T xin = rows(X, n0, n);
    for (int i = 0; i < 4; ++i) {
      T yv = rows(pre[i], n0, n);
      if (i == 0) {
        // d_conv_0 + LeakyReLU (fused where the thin kernel runs the layer); the tangent rows: the convolution alone, then the
        // LeakyReLU's linearisation at the primal (rows [0,B) of the same tensor)
        if (ne > 0) {
          T xe = rows(X, n0, ne), ye = rows(pre[0], n0, ne);
          // bf16 mode: layer 0's output -- the largest activation of the sub-step -- as its plane alone: read by d_conv_1 and its
          // filter gradient (planes) and, for the LeakyReLU's backward, by its sign (ConvArgs::epi_ref_bf, launch_act_bwd_bf)
          c.conv_plane_only_next = planes && out[0]->d_bf != nullptr;
          const int r = conv_into(c, xe, w[0], ye, 4, 2, ACT_LRELU);
          if (ye->d_plane_only) pre[0]->d_plane_only = true;
          if (!(r & 1)) launch_act_fwd(ye->d, ye->d, ye->numel(), ACT_LRELU, c.stream);
          if (ye->d_bf && !(r & 2)) launch_cvt_bf16(ye->d, ye->d_bf, ye->numel(), c.stream);   // (plane rows the kernel did not store)
        }
        if (nt > 0) {
          T xt = rows(X, n0 + ne, nt), yt = rows(pre[0], n0 + ne, nt);
          conv_into(c, xt, w[0], yt, 4, 2, 0);
          if (pre[0]->d_plane_only) {     // t_out = act'(out_primal) * t: mask from the primal rows' plane, the result as its plane
            launch_act_bwd_bf(pre[0]->d_bf, yt->d, nullptr, yt->d_bf, yt->numel(), ACT_LRELU, c.stream);
          } else {
            launch_act_bwd(pre[0]->d, yt->d, yt->d, 0, yt->numel(), ACT_LRELU, c.stream);
            if (yt->d_bf) launch_cvt_bf16(yt->d, yt->d_bf, yt->numel(), c.stream);
          }
        }
      } else {
        conv_into(c, xin, w[i], yv, 4, 2, 0);
        const int HW = pre[i]->h * pre[i]->w, C = ch[i + 1];
        const size_t img = (size_t)HW * C;
        if (ne > 0)
          launch_in_fwd(yv->d, out[i]->d_plane_only ? nullptr : out[i]->d + (size_t)n0 * img, st[i]->d + (size_t)n0 * 2 * C, ne, HW, C,
                        ACT_LRELU, c.stream, 0, 0, nullptr, nullptr, out[i]->d_bf ? out[i]->d_bf + (size_t)n0 * img : nullptr,
                        in_chunk_scratch(c, ne, HW, C));
        if (nt > 0)   // the tangent rows: linearised at the primal rows [0,B) (x, out, statistics)
          launch_in_tangent_fwd(pre[i]->d, out[i]->d, st[i]->d, yv->d + (size_t)ne * img,
                                out[i]->d_plane_only ? nullptr : out[i]->d + (size_t)(n0 + ne) * img, nt, HW, C, ACT_LRELU, c.stream,
                                out[i]->d_bf ? out[i]->d_bf + (size_t)(n0 + ne) * img : nullptr, in_chunk_scratch(c, nt, HW, C));
      }
      xin = rows(out[i], n0, n);
    }
    // d_linear_5: bias on the evaluated rows only (the tangent of a constant is zero)
    const size_t feat = (size_t)K;
    if (ne > 0) launch_gemv_fwd(out[3]->d + (size_t)n0 * feat, lw->w, lb->w, dval + n0, ne, K, c.stream);
    if (nt > 0) launch_gemv_fwd(out[3]->d + (size_t)(n0 + ne) * feat, lw->w, nullptr, dval + n0 + ne, nt, K, c.stream);
  };
  // norm + activation backward of layer i for rows [n0, n0 + n) of the evaluated points: gout -> gpre
  static const bool gpo_on = dbg_env("EG_NO_PLANE_ONLY_GRAD") == nullptr;
  const int gpo = gpo_on ? GRAD_PLANE_ONLY : 0;
  // (gy0: the rows of out[i]'s GRADIENT that hold dL/d(out) of these points -- phase 1 keeps the first-order gradient of the
  //  x_hat rows in the tangent rows, see above)
  auto norm_bwd = [&](int i, int n0, int n, int acc, int gy0) {
    const int HW = pre[i]->h * pre[i]->w, C = ch[i + 1];
    const size_t img = (size_t)HW * C, o = (size_t)n0 * img;
    // (gpo: the fp32 gradient of pre[i] is not stored -- both gradient convolutions of layer i read the plane, Tensor::g_plane_only)
    launch_in_bwd(pre[i]->d + o, out[i]->d + o, out[i]->g + (size_t)gy0 * img, st[i]->d + (size_t)n0 * 2 * C, pre[i]->g + o, acc | (gpl[i] ? gpo : 0), n, HW, C,
                  ACT_LRELU, c.stream, gpl[i] ? gpl[i] + o : nullptr, nullptr, nullptr, in_chunk_scratch(c, n, HW, C));
    pre[i]->g_bf = gpl[i];            // (rows [n0, n0 + n) of the plane are current; view_rows hands them on)
    pre[i]->g_bf_stream = nullptr;
    pre[i]->g_plane_only = gpl[i] && gpo;
  };
  // ---- phase 1: x_hat, and the first-order gradient g = d/dx_hat sum_b [sigmoid(D) + D] (quirk N10) ----
  // (data of the x_hat rows [0,B); the gradients dL/d(out_i) go to the tangent rows [3B,4B) of out[i]->g, where the reverse
  //  sweep needs them as the cotangents of the tangent)
  auto grad_rows = [&](T t, int data0, int grad0, int n) {   // rows [data0, data0 + n) of t whose gradient lives in rows [grad0, grad0 + n)
    T v = rows(t, data0, n);
    v->g = t->g + (size_t)grad0 * (t->numel() / t->n);
    v->g_bf = nullptr;
    v->g_plane_only = false;
    return v;
  };
  forward(0, B);
  {
    launch_gp_head(dval, nullptr, dgrad, nullptr, B, 0, c.stream);
    T o3 = grad_rows(out[3], 0, 3 * B, B);
    launch_gemv_bwd_x(dgrad, lw->w, o3->g, 0, B, K, c.stream);
    o3->g_set = true;
    bool act0_done = false;
    for (int i = 3; i >= 1; --i) {
      norm_bwd(i, 0, B, 0, 3 * B);
      T yv = rows(pre[i], 0, B), xv = grad_rows(out[i - 1], 0, 3 * B, B);
      yv->g_set = true;
      if (i == 1) {   // layer 0's LeakyReLU backward rides in this launch's epilogue where the kernel that runs has one
        xv->fused_act = ACT_LRELU;
        xv->g_inplace = true;
      }
      conv_backward_x(c, xv, yv, w[i], 4, 2);
      if (i == 1) act0_done = xv->g_act_done;
    }
    T y0 = grad_rows(pre[0], 0, 3 * B, B), x0 = rows(X, 0, B);
    if (!act0_done) {
      tensor_need_f32(c, y0);
      launch_act_bwd(y0->d, y0->g, y0->g, 0, y0->numel(), ACT_LRELU, c.stream);   // in place: g *= act'(out)
    }
    y0->g_set = true;
    x0->needs_grad = true;
    conv_backward_x(c, x0, y0, w[0], 4, 2);
  }
  float* norms = c.arena.allocf(B);
  launch_gp_v(X->g, X->d + (size_t)3 * B * per, norms, c.d_losses + slot_gp, B, per, cfg.lambda_gp, c.stream);   // v -> the tangent rows of X
  // ---- phase 2: real | fake | tangent in one pass ----
  forward(B, 3 * B);
  launch_mean_diff(dval + 2 * B, dval + B, c.d_losses + slot_d, B, c.stream);
  // ---- the reverse sweep: filter gradients over all 4B rows, input gradients over the 3B evaluated rows ----
  launch_gp_head(dval, dval + 3 * B, dgrad, dgrad + 3 * B, B, 1, c.stream);   // seeds of d(x_hat) and of the tangent
  launch_fill(dgrad + B, -1.f / B, B, c.stream);
  launch_fill(dgrad + 2 * B, 1.f / B, B, c.stream);
  linear1_backward_w(c, out[3]->d, dgrad, lw, lb, 3 * B, R, K);   // (the tangent rows carry no bias)
  launch_gemv_bwd_x(dgrad, lw->w, out[3]->g, 0, 3 * B, K, c.stream);   // (the tangent rows of out[3]->g: phase 1's s(d) x Matrix, the same product)
  bool act0_all = false;
  for (int i = 3; i >= 1; --i) {
    const int HW = pre[i]->h * pre[i]->w, C = ch[i + 1];
    const size_t img = (size_t)HW * C, ot = (size_t)3 * B * img;
    norm_bwd(i, B, 2 * B, 0, B);                               // real | fake
    // the tangent rows: reverse of the linearisation -- gradient w.r.t. the tangent (= phase 1's delta(pre_i), formed again by the
    // same kernel that forms the term below) AND w.r.t. the primal pre-norm rows [0,B)
    launch_in_tangent_bwd(pre[i]->d, out[i]->d, st[i]->d, pre[i]->d + ot, out[i]->g + ot, pre[i]->g + ot, gpl[i] ? gpo : 0, pre[i]->g, 0, B, HW, C,
                          ACT_LRELU, c.stream, gpl[i] ? gpl[i] + ot : nullptr, in_chunk_scratch(c, B, HW, C));
    norm_bwd(i, 0, B, 1, 0);                                   // the primal rows add their own first-order term
    pre[i]->g_set = true;
    conv_backward_w(c, out[i - 1], pre[i], w[i], 4, 2);
    // input gradient towards the evaluated rows only (the tangent rows of out[i-1]->g hold phase 1's gradient)
    T yv = rows(pre[i], 0, 3 * B), xv = rows(out[i - 1], 0, 3 * B);
    yv->g_set = true;
    xv->needs_grad = true;
    if (i == 1) {   // layer 0's LeakyReLU backward in the epilogue (every row by its own output)
      xv->fused_act = ACT_LRELU;
      xv->g_inplace = true;
    }
    conv_backward_x(c, xv, yv, w[i], 4, 2);
    if (i == 1) act0_all = xv->g_act_done;
  }
  if (!act0_all) {   // layer 0: LeakyReLU backward on the evaluated rows (the tangent rows had theirs in phase 1, by the primal's mask)
    const size_t img = pre[0]->numel() / R;
    tensor_need_f32(c, pre[0]);
    launch_act_bwd(pre[0]->d, pre[0]->g, pre[0]->g, 0, (size_t)3 * B * img, ACT_LRELU, c.stream);
  }
  {
    pre[0]->g_set = true;
    pre[0]->g_bf = nullptr;
    pre[0]->g_plane_only = false;
    conv_backward_w(c, X, pre[0], w[0], 4, 2);
  }
  flush_deferred(c);
  if (apply) apply_group(c, group);
  set_want_grad(c, group, false);
}

// The stacked form runs every InstanceNorm discriminator sub-step (its coverage is the oracle: every discriminator sub-step test
// of tests/test_step_gpu.py / test_tight_gpu.py / test_x3_gpu.py / test_bf16_gpu.py).  The tape-driven form at 7F - 2F0 runs the
// sub-steps under D_norm = 'batch' (tests/test_step_gpu.py::test_substeps_batch_norm_flags against the oracle) and, with EG_D_TAPE=1
// in the experiments build (dbg_env is compiled out of the shipped library), the InstanceNorm ones for in-box A/B runs.
static void d_substep(Ctx& c, int group, T real, T fake, const float* alpha_dev, int slot_d, int slot_gp, bool apply) {
  static const bool tape_form = dbg_env("EG_D_TAPE") != nullptr;
  // (D_norm = 'batch': the tape form -- the stacked sweep's row classes are InstanceNorm's, and BatchNorm couples the rows of a call)
  if (tape_form || c.cfg.d_bn || c.cfg.d_nonorm) disc_substep(c, group, real, fake, alpha_dev, slot_d, slot_gp, apply);
  else disc_substep_stacked(c, group, real, fake, alpha_dev, slot_d, slot_gp, apply);
}

// G1(z) / G2(z) for the discriminator sub-steps: evaluated once per iteration and shared by sub-steps 1-3 (the
// reference re-evaluates them in each sess.run; the generators are frozen until sub-step 5, so the values are the
// same -- SURVEY 7.2).  The BatchNorm moving statistics still advance once per reference evaluation.
static T generator_output_nograd(Ctx& c, int group, T zin, int bn_t) {
  const Config& cfg = c.cfg;
  const int i = group == G_G1 ? 0 : 1;
  const std::string p = kGroupPrefix[group];
  const int C = cfg.gf * 8, R = cfg.B * (cfg.H / 16) * (cfg.W / 2 / 16);
  const int bch[3] = {cfg.gf * 4, cfg.gf * 2, cfg.gf};   // G_norm = 'batch': the block norms of g_dconv_1 .. 3
  if (c.gen_cache_valid[i]) {
    if (bn_t > 0) {
      launch_bn_moving_update(P(c, p + "/batch_norm/BatchNorm/moving_mean")->w,
                              P(c, p + "/batch_norm/BatchNorm/moving_variance")->w, c.d_gen_bn[i], C, R, 0.9f, bn_t,
                              c.stream);
      for (int j = 0; cfg.g_bn && j < 3; ++j) {
        const std::string b = p + "/g_dconv_" + std::to_string(j + 1) + "/batch_norm/BatchNorm/";
        launch_bn_moving_update(P(c, b + "moving_mean")->w, P(c, b + "moving_variance")->w, c.d_gen_bn_blk[i][j], bch[j],
                                R << (2 * (j + 1)), 0.9f, bn_t, c.stream);
      }
    }
    return wrap_tensor(c, c.d_gen_cache[i], cfg.B, cfg.H, cfg.W / 2, cfg.c_dim);
  }
  T st = nullptr;
  T out = generator_forward(c, group, zin, bn_t, &st, c.d_gen_cache[i]);   // (forward-only: the last layer stores tanh(y) straight into the cache)
  if (out->d != c.d_gen_cache[i]) launch_copy(out->d, c.d_gen_cache[i], out->numel(), c.stream);
  launch_copy(st->d, c.d_gen_bn[i], (size_t)2 * C, c.stream);
  for (int j = 0; cfg.g_bn && j < 3; ++j) launch_copy(c.gen_blk_stats[j]->d, c.d_gen_bn_blk[i][j], (size_t)2 * bch[j], c.stream);
  c.gen_cache_valid[i] = true;
  return out;
}

static T right_half(Ctx& c, T images) { return slice_w(c, images, c.cfg.W / 2, c.cfg.W); }
static T left_half(Ctx& c, T images) { return slice_w(c, images, 0, c.cfg.W / 2); }

void run_substep(Ctx& c, int k, bool apply, const SubOpts* opts) {
  const Config& cfg = c.cfg;
  const int B = cfg.B;
  EG_CHECK(k >= 1 && k <= 7, "sub-step index 1..7");
  if (k == 4 && !cfg.multiclasses()) return;  // d_optim2 only registered with multiclasses (edgegan.py:115-116)
  if ((k == 2 && !cfg.use_image_dis) || (k == 3 && !cfg.use_edge_dis)) return;   // register_optim_if(..., use_*_discriminator), edgegan.py:111-114
  {  // weights (and flat gradient buffers) this sub-step touches must have their pending update applied
    static const unsigned reads[8] = {0,
                                      1u << G_G1 | 1u << G_G2 | 1u << G_D,
                                      1u << G_G2 | 1u << G_DP2,
                                      1u << G_G1 | 1u << G_DP3,
                                      1u << G_D2,
                                      1u << G_G1 | 1u << G_G2 | 1u << G_D | 1u << G_DP2 | 1u << G_DP3 | 1u << G_D2,
                                      1u << G_G1 | 1u << G_E,
                                      1u << G_G1 | 1u << G_G2 | 1u << G_D | 1u << G_DP2 | 1u << G_DP3 | 1u << G_D2};
    unsigned mask = reads[k];
    if (opts && opts->wait_for_group)   // the discriminators' / classifier's pending updates are awaited at first use
      mask &= ~(1u << G_D | 1u << G_DP2 | 1u << G_DP3 | 1u << G_D2);
    comm_wait_groups(c, mask);
  }
  reset_step(c);
  for (int g = 0; g < G_COUNT; ++g) set_want_grad(c, g, false);
  T images = wrap_tensor(c, c.d_images, B, cfg.H, cfg.W, cfg.c_dim);
  // contrib batch_norm updates its moving statistics whenever a generator forward is evaluated in a
  // training run (updates_collections=None); gradient-only probes (apply=false) leave them alone.
  const int bn_t = apply ? 1 : 0;
  c.bn_times_block = bn_t;   // (E_norm / D_norm = 'batch': a block norm advances its moving statistics once per evaluation)
  const int bn1 = (opts && opts->bn_times_g1 >= 0) ? opts->bn_times_g1 : bn_t;   // G1 / G2 moving-statistics updates
  const int bn2 = (opts && opts->bn_times_g2 >= 0) ? opts->bn_times_g2 : bn_t;
  auto await_group = [&](int g) {   // see SubOpts::wait_for_group
    if (!(opts && opts->wait_for_group)) return;
    // (a lane does not wait for its OWN event: stream order already is that dependency -- and a stream capture of the iteration
    //  does not survive one: hipStreamEndCapture crashes, docs/experiments.md "the iteration as a HIP graph")
    if (opts->wait_for_group[g] && opts->wait_for_group[g] != c.lanes[c.cur_lane].done)
      EG_HIP(hipStreamWaitEvent(c.stream, opts->wait_for_group[g], 0));
    comm_wait_groups(c, 1u << g);
  };
  auto before_apply = [&]() {
    if (opts && opts->wait_before_apply) EG_HIP(hipStreamWaitEvent(c.stream, opts->wait_before_apply, 0));
  };
  switch (k) {
    case 1: {  // d_optim: joint discriminator
      T zin = make_zin(c);
      T edge = generator_output_nograd(c, G_G1, zin, bn1);
      T image = generator_output_nograd(c, G_G2, zin, bn2);
      if (opts && opts->record_after_gen) EG_HIP(hipEventRecord(opts->record_after_gen, c.stream));
      T joint = concat_w(c, edge, image);
      d_substep(c, G_D, images, joint, c.d_alpha, L_JOINT_D, L_JOINT_GP, apply);
    } break;
    case 2: {  // d_optim_patch2: image discriminator on bicubic-resized right halves
      T zin = make_zin(c);
      T image = generator_output_nograd(c, G_G2, zin, bn2);
      T r_real = resize(c, right_half(c, images), cfg.image_dis);
      T r_fake = resize(c, image, cfg.image_dis);
      d_substep(c, G_DP2, r_real, r_fake, c.d_alpha + B, L_IMAGE_D, L_IMAGE_GP, apply);
    } break;
    case 3: {  // d_optim_patch3: edge discriminator
      T zin = make_zin(c);
      T edge = generator_output_nograd(c, G_G1, zin, bn1);
      T r_real = resize(c, left_half(c, images), cfg.edge_dis);
      T r_fake = resize(c, edge, cfg.edge_dis);
      d_substep(c, G_DP3, r_real, r_fake, c.d_alpha + 2 * B, L_EDGE_D, L_EDGE_GP, apply);
    } break;
    case 4: {  // d_optim2: classifier focal loss on the real right half (edgegan.py:238-239,325-330)
      set_want_grad(c, G_D2, true);
      T logits = classifier_forward(c, right_half(c, images));
      seed_grad(c, logits);
      launch_acgan_loss(logits->d, c.d_z + cfg.z_dim, cfg.z_dim + 1, c.d_losses + L_CLS_D, logits->g, B,
                        cfg.num_classes, 0, c.scratch, c.stream);
      c.tape.backward();
      if (apply) apply_group(c, G_D2);
    } break;
    case 5:
    case 7: {  // g_optim_u / g_optim_b: both generator minimize ops in one run (quirk N9)
      set_want_grad(c, G_G1, true);
      set_want_grad(c, G_G2, true);
      T zin = make_zin(c);
      // G2 first: the classifier branch -- the longest of the four critics, and with sub-step 4 in front of it the
      // iteration's critical chain -- reads G2(z) only, so with branch lanes it forks off right behind G2's forward
      // (ev_cls) while G1's forward, the concat and the resizes are still being computed on lane 0
      const bool bl = opts && opts->branch_lanes && c.cur_lane == 0;
      const size_t mg2 = c.tape.ops.size();
      T image = generator_forward(c, G_G2, zin, bn2);
      if (bl) EG_HIP(hipEventRecord(c.ev_cls, c.stream));
      const size_t mg1 = c.tape.ops.size();
      T edge = generator_forward(c, G_G1, zin, bn1);
      const size_t mcat = c.tape.ops.size();   // concat: entries mcat (edge's part), mcat + 1 (image's part)
      T joint = concat_w(c, edge, image);
      // (a patch discriminator that is switched off contributes neither a branch nor a loss term: edge_dis_gloss /
      //  image_dis_gloss = 0, edgegan.py:300-301,310-311; its tape ranges below are empty)
      const size_t mre = c.tape.ops.size();
      T r_edge = cfg.use_edge_dis ? resize(c, edge, cfg.edge_dis) : nullptr;
      const size_t mri = c.tape.ops.size();
      T r_image = cfg.use_image_dis ? resize(c, image, cfg.image_dis) : nullptr;
      // The four critics are independent given the generator outputs (each writes the gradient of its OWN input tensor;
      // `image`'s gradient gets the classifier's contributions first, then -- after the join -- the resize's and the
      // concat's, the order of the single-stream sweep).  With branch lanes each runs forward and backward on the lane that
      // trained it, directly behind that sub-step.
      const size_t m0 = c.tape.ops.size();
      if (bl) EG_HIP(hipEventRecord(c.ev_branch, c.stream));
      auto enter = [&](int lane, hipEvent_t after) {
        if (!bl || lane == 0) return;
        use_lane(c, lane);
        c.arena.reset();   // (this lane's previous sub-step is fully enqueued; stream order protects its memory)
        EG_HIP(hipStreamWaitEvent(c.stream, after, 0));
      };
      await_group(G_D);
      T dj = disc_forward(c, G_D, joint, nullptr, nullptr);
      launch_mean_neg(dj->d, c.d_losses + L_JOINT_G, B, c.stream);
      seed_grad(c, dj);
      launch_fill(dj->g, -cfg.joint_dweight / B, B, c.stream);
      const size_t m1 = c.tape.ops.size();
      if (cfg.use_edge_dis) {
        enter(2, c.ev_branch);
        await_group(G_DP3);
        T de = disc_forward(c, G_DP3, r_edge, nullptr, nullptr);
        launch_mean_neg(de->d, c.d_losses + L_EDGE_G, B, c.stream);
        seed_grad(c, de);
        launch_fill(de->g, -cfg.edge_dweight / B, B, c.stream);
      }
      const size_t m2 = c.tape.ops.size();
      if (cfg.use_image_dis) {
        enter(1, c.ev_branch);
        await_group(G_DP2);
        T di = disc_forward(c, G_DP2, r_image, nullptr, nullptr);
        launch_mean_neg(di->d, c.d_losses + L_IMAGE_G, B, c.stream);
        seed_grad(c, di);
        launch_fill(di->g, -cfg.image_dweight / B, B, c.stream);
      }
      const size_t m3 = c.tape.ops.size();
      if (cfg.multiclasses()) {
        enter(3, c.ev_cls);
        await_group(G_D2);
        T lf = classifier_forward(c, image);
        seed_grad(c, lf);
        launch_acgan_loss(lf->d, c.d_z + cfg.z_dim, cfg.z_dim + 1, c.d_losses + L_CLS_G, lf->g, B, cfg.num_classes,
                          1, c.scratch, c.stream);
      }
      const size_t m4 = c.tape.ops.size();
      if (!bl) {
        c.tape.backward();
      } else {
        auto sweep = [&](int lane, size_t lo, size_t hi) {
          use_lane(c, lane);
          c.tape.backward_range(lo, hi);
          if (lane != 0) EG_HIP(hipEventRecord(c.lanes[lane].done, c.stream));
        };
        if (m4 > m3) sweep(3, m3, m4);
        // resize() is the identity when H == W/2 == dis_size: r_image IS image, so the image discriminator's input
        // gradient accumulates into the buffer the classifier's sweep (lane 3) writes first -- order lane 1 behind it
        if (m4 > m3 && r_image == image) {
          use_lane(c, 1);
          EG_HIP(hipStreamWaitEvent(c.stream, c.lanes[3].done, 0));
        }
        sweep(1, m2, m3);
        sweep(2, m1, m2);
        sweep(0, m0, m1);
        // G1's side needs the joint (lane 0) and the edge (lane 2) discriminators only: d(edge) = resize backward, then the
        // concat's edge columns -- the per-buffer order of the single-stream sweep -- and G1's backward chain run while
        // the classifier branch is still busy.  G2's side follows once the image discriminator and the classifier are in.
        EG_HIP(hipStreamWaitEvent(c.stream, c.lanes[2].done, 0));
        c.tape.backward_range(mre, mri);          // resize(edge)
        c.tape.backward_range(mcat, mcat + 1);    // concat: edge columns
        c.tape.backward_range(mg1, mcat);         // G1
        EG_HIP(hipStreamWaitEvent(c.stream, c.lanes[1].done, 0));
        if (m4 > m3) EG_HIP(hipStreamWaitEvent(c.stream, c.lanes[3].done, 0));
        c.tape.backward_range(mri, m0);           // resize(image)
        c.tape.backward_range(mcat + 1, mre);     // concat: image columns
        c.tape.backward_range(mg2, mg1);          // G2
        c.tape.backward_range(0, mg2);
      }
      if (apply) {
        before_apply();
        apply_generators(c);
      }
    } break;
    case 6: {  // e_optim: L1 latent reconstruction through G1 (edgegan.py:275,337-342)
      set_want_grad(c, G_E, true);
      T zin = make_zin(c);
      T edge = generator_forward(c, G_G1, zin, bn1);
      if (opts && opts->record_after_g1_fwd) EG_HIP(hipEventRecord(opts->record_after_g1_fwd, c.stream));
      T zr = encoder_forward(c, edge, c.d_eps);
      seed_grad(c, zr);
      launch_l1(c.d_z, cfg.multiclasses() ? cfg.z_dim + 1 : cfg.z_dim, zr->d, c.d_losses + L_ZL, zr->g, B, cfg.z_dim,
                cfg.zl_weight, c.stream);
      c.tape.backward();
      if (apply) apply_group(c, G_E);
    } break;
  }
  if (apply && (k == 4)) c.sn_valid = false;
}

// Transposed copies of a group's conv filters, made on the CURRENT stream (used before two lanes read the same
// generator: the lazily built cache would otherwise be filled on one lane's stream and read on the other's).
static void warm_transposes(Ctx& c, int group) {
  for (Param* p : c.groups[group].params) {
    if (!p->cache_T || p->shape.size() != 4 || p->wT_valid) continue;
    if (!p->wT) EG_HIP(hipMalloc(&p->wT, p->numel * sizeof(float)));
    launch_tap_transpose(p->w, p->wT, p->shape[0] * p->shape[1], p->shape[2], p->shape[3], nullptr, c.stream);
    p->wT_valid = true;
  }
  for (Param* p : c.groups[group].params) {   // the generator's last deconv runs phase-merged (ops.cpp: merged_filter)
    if (!p->cache_T || p->shape.size() != 4 || p->shape[2] != 3 || p->shape[3] % 16 != 0 || p->wM_valid) continue;
    const size_t n = (size_t)9 * p->shape[3] * 12;
    if (!p->wM) EG_HIP(hipMalloc(&p->wM, n * sizeof(float)));
    launch_merge_phase_filter(p->w, p->wM, p->shape[0], p->shape[3], c.stream);
    p->wM_valid = true;
  }
}

// One iteration = the 7 sub-steps of update_model in the reference's order.  Their data dependencies are sparser
// than that order: 1-4 train four networks with disjoint variables on inputs that nothing in 1-4 changes (the
// generators are frozen until 5), 5 needs all of them, 6 (encoder, reads G1) and 7 (generators) are independent
// except that 7's update of G1 must come after 6's G1 forward.  train_step enqueues {1,2,3,4} on four lanes, 5 on
// lane 0, {6,7} on two lanes; every kernel still sees exactly the inputs of the sequential schedule, so the results
// are bitwise those of run_substep(1..7) (tests/test_step_gpu.py).  Measured gain: see DESIGN.md section 6.
void train_step(Ctx& c) {
  const bool conc = c.lanes_on && !c.prof_on;
  if (!conc) {
    for (int k = 1; k <= 7; ++k) run_substep(c, k, true);
    ++c.step_count;
    return;
  }
  const bool cls = c.cfg.multiclasses();
  static const bool early_cls_probe = dbg_env("EG_EARLY_CLS") != nullptr;   // timing probe: lane 3 does not wait for lane 0's tail
  auto fork = [&](int nl) {   // lanes 1..nl-1 start after everything already enqueued on lane 0
    EG_HIP(hipEventRecord(c.ev_fork, c.lanes[0].stream));
    for (int i = 1; i < nl; ++i)
      if (!(early_cls_probe && i == 3 && nl == 4)) EG_HIP(hipStreamWaitEvent(c.lanes[i].stream, c.ev_fork, 0));
  };
  use_lane(c, 0);
  const bool conc_a = c.lanes_mask & 1, conc_c = c.lanes_mask & 2;
  // ---- phase A: the four discriminator-side sub-steps ------------------------------------------------------
  if (!conc_a) {
    for (int k = 1; k <= 4; ++k) run_substep(c, k, true);
  } else {
  fork(4);
  {
    // the classifier sub-step is the longest of the four and the one whose gradient exchange (69 MB under data
    // parallelism) sub-step 5 waits for: enqueue it first so its all-reduce runs under the discriminator sub-steps
    // EG_PIPE_PROBE (experiments build; timing only, the results are wrong): the classifier sub-step on a fifth lane that nothing
    // waits for before the end of the iteration -- what a software pipeline (the NEXT batch's classifier gradients under sub-steps
    // 5 - 7) would overlap; docs/experiments.md
    static const bool pipe_probe = dbg_env("EG_PIPE_PROBE") != nullptr;
    if (cls) {
      if (pipe_probe) {
        ensure_lane(c, 4);
        EG_HIP(hipStreamWaitEvent(c.lanes[4].stream, c.ev_fork, 0));
      }
      use_lane(c, pipe_probe ? 4 : 3);
      run_substep(c, 4, true);
      use_lane(c, 0);
    }
    // sub-steps 1, 2, 3 share one evaluation of G1(z), G2(z); each reference evaluation advances the BatchNorm
    // moving statistics once (G1: sub-steps 1 and 3, G2: 1 and 2) -- applied here as two updates by sub-step 1
    SubOpts o1;
    o1.bn_times_g1 = 2;
    o1.bn_times_g2 = 2;
    o1.record_after_gen = c.ev_gen;
    run_substep(c, 1, true, &o1);
    SubOpts o23;
    o23.bn_times_g1 = 0;
    o23.bn_times_g2 = 0;
    use_lane(c, 1);
    EG_HIP(hipStreamWaitEvent(c.stream, c.ev_gen, 0));
    run_substep(c, 2, true, &o23);
    use_lane(c, 2);
    EG_HIP(hipStreamWaitEvent(c.stream, c.ev_gen, 0));
    run_substep(c, 3, true, &o23);
  }
  // no join: sub-step 5 follows sub-step 1 on lane 0 and waits for lanes 1-3 only where it first uses the network
  // each of them trained, so its generator forwards and the joint-discriminator pass overlap the tail of phase A
  use_lane(c, 0);
  for (int i = 1; i < 4; ++i) EG_HIP(hipEventRecord(c.lanes[i].done, c.lanes[i].stream));
  }
  // ---- phase B: generators against all four ----------------------------------------------------------------
  if (conc_a) {
    hipEvent_t ev[G_COUNT] = {};
    ev[G_DP2] = c.lanes[1].done;
    ev[G_DP3] = c.lanes[2].done;
    ev[G_D2] = c.lanes[3].done;
    SubOpts o5;
    o5.wait_for_group = ev;
    o5.branch_lanes = (c.lanes_mask & 4) != 0;
    run_substep(c, 5, true, &o5);
  } else {
    SubOpts o5;
    o5.branch_lanes = (c.lanes_mask & 4) != 0;
    run_substep(c, 5, true, &o5);
  }
  // ---- phase C: encoder (reads G1) next to the second generator step (updates G1 at its end) -----------------
  if (!conc_c) {
    run_substep(c, 6, true);
    SubOpts o7;
    o7.branch_lanes = (c.lanes_mask & 4) != 0;
    run_substep(c, 7, true, &o7);
    ++c.step_count;
    return;
  }
  comm_wait_groups(c, 1u << G_G1);   // G1's pending update (data parallel) must be in before its filters are read
  warm_transposes(c, G_G1);
  // (lane 1 -- the lane whose hardware queue also hosts the leaf streams -- measured best: lanes 2 / 3 -1.9 % / -1.5 %,
  //  profiles/r03v_ab_substep6_lane*.txt)
  static const int l6 = dbg_env("EG_SS6_LANE") ? atoi(dbg_env("EG_SS6_LANE")) : 1;
  EG_HIP(hipEventRecord(c.ev_fork, c.lanes[0].stream));
  EG_HIP(hipStreamWaitEvent(c.lanes[l6].stream, c.ev_fork, 0));
  {
    SubOpts o6;
    o6.bn_times_g1 = 0;              // sub-step 7 evaluates the same G1 forward: it applies both updates
    o6.record_after_g1_fwd = c.ev_g1fwd;
    use_lane(c, l6);
    run_substep(c, 6, true, &o6);
    SubOpts o7;
    o7.bn_times_g1 = 2;
    o7.bn_times_g2 = 1;
    o7.wait_before_apply = c.ev_g1fwd;
    o7.branch_lanes = (c.lanes_mask & 4) != 0;
    use_lane(c, 0);
    run_substep(c, 7, true, &o7);
  }
  use_lane(c, 0);
  EG_HIP(hipEventRecord(c.lanes[l6].done, c.lanes[l6].stream));
  EG_HIP(hipStreamWaitEvent(c.lanes[0].stream, c.lanes[l6].done, 0));
  if (c.lanes[4].stream && dbg_env("EG_PIPE_PROBE")) {   // (probe: join the fifth lane)
    EG_HIP(hipEventRecord(c.lanes[4].done, c.lanes[4].stream));
    EG_HIP(hipStreamWaitEvent(c.lanes[0].stream, c.lanes[4].done, 0));
  }
  ++c.step_count;
}

// Forward-only evaluation of every loss with the CURRENT weights -- what the reference's observe-only runs compute
// after update_model (sess.run(d_sum), sess.run(g_sum) and the five evaluate() calls of edgegan.py:427-433,464-480):
// fresh random_blend draws (alpha), a fresh encoder noise scalar, and the generators' BatchNorm moving statistics
// advanced once per run that evaluates them (bn_updates; 6 mirrors the reference loop: d_sum, g_sum, joint_dis_dloss,
// one patch loss, edge_gloss, image_gloss).  The gradient penalty still needs the input gradient of each
// discriminator (one first-order reverse sweep, no parameter gradients).
void eval_losses(Ctx& c, const float* alpha_dev, const float* eps_dev, int bn_updates, EvalResult* r) {
  const Config& cfg = c.cfg;
  const int B = cfg.B;
  comm_wait_all(c);
  reset_step(c);
  for (int g = 0; g < G_COUNT; ++g) set_want_grad(c, g, false);
  T images = wrap_tensor(c, c.d_images, B, cfg.H, cfg.W, cfg.c_dim);
  T zin = make_zin(c);
  c.bn_times_block = bn_updates;
  T edge = generator_forward(c, G_G1, zin, bn_updates);
  T image = generator_forward(c, G_G2, zin, bn_updates);
  T joint = concat_w(c, edge, image);
  T r_pic = nullptr, r_img = nullptr, r_edges = nullptr, r_edge_out = nullptr;
  if (cfg.use_image_dis) r_pic = resize(c, right_half(c, images), cfg.image_dis), r_img = resize(c, image, cfg.image_dis);
  if (cfg.use_edge_dis) r_edges = resize(c, left_half(c, images), cfg.edge_dis), r_edge_out = resize(c, edge, cfg.edge_dis);
  float* L = c.d_eval;
  struct Branch {
    int group;
    T real, fake;
    int slot_d, slot_gp, slot_g;
  } br[3] = {{G_D, images, joint, L_JOINT_D, L_JOINT_GP, L_JOINT_G},
             {G_DP2, r_pic, r_img, L_IMAGE_D, L_IMAGE_GP, L_IMAGE_G},
             {G_DP3, r_edges, r_edge_out, L_EDGE_D, L_EDGE_GP, L_EDGE_G}};
  for (int i = 0; i < 3; ++i) {
    const Branch& b = br[i];
    if (!b.real) continue;   // a patch discriminator that is switched off: its losses stay 0 (edgegan.py:300-301,310-311)
    T xhat = new_tensor(c, b.real->n, b.real->h, b.real->w, b.real->c, true);
    const size_t per = xhat->numel() / B;
    launch_blend(b.real->d, b.fake->d, alpha_dev + (size_t)i * B, xhat->d, B, per, c.stream);
    const size_t mark = c.tape.ops.size();
    T d_hat = disc_forward(c, b.group, xhat, nullptr, nullptr);
    seed_grad(c, d_hat);
    launch_gp_head(d_hat->d, nullptr, d_hat->g, nullptr, B, 0, c.stream);
    c.param_grad_enabled = false;
    c.tape.backward_range(mark, c.tape.ops.size());
    c.param_grad_enabled = true;
    float* v = c.arena.allocf(xhat->numel());
    float* norms = c.arena.allocf(B);
    launch_gp_v(xhat->g, v, norms, L + b.slot_gp, B, per, cfg.lambda_gp, c.stream);
    c.tape.ops.resize(mark);
    reset_grads(c);
    if (cfg.d_bn) {   // (BatchNorm statistics are per call: real and fake are two evaluations)
      T d_real = disc_forward(c, b.group, b.real, nullptr, nullptr), d_fake = disc_forward(c, b.group, b.fake, nullptr, nullptr);
      launch_mean_diff(d_fake->d, d_real->d, L + b.slot_d, B, c.stream);
      launch_mean_neg(d_fake->d, L + b.slot_g, B, c.stream);
      launch_copy(d_real->d, c.d_eval + 16 + (size_t)i * 2 * B, (size_t)B, c.stream);   // logits: real [B] | fake [B]
      launch_copy(d_fake->d, c.d_eval + 16 + (size_t)i * 2 * B + B, (size_t)B, c.stream);
      continue;
    }
    T rf = new_tensor(c, 2 * B, b.real->h, b.real->w, b.real->c, false);
    launch_copy(b.real->d, rf->d, b.real->numel(), c.stream);
    launch_copy(b.fake->d, rf->d + b.real->numel(), b.fake->numel(), c.stream);
    T d_rf = disc_forward(c, b.group, rf, nullptr, nullptr);
    launch_mean_diff(d_rf->d + B, d_rf->d, L + b.slot_d, B, c.stream);
    launch_mean_neg(d_rf->d + B, L + b.slot_g, B, c.stream);
    launch_copy(d_rf->d, c.d_eval + 16 + (size_t)i * 2 * B, (size_t)2 * B, c.stream);   // logits: real [B] | fake [B]
  }
  if (cfg.multiclasses()) {
    T lr = classifier_forward(c, right_half(c, images));
    launch_acgan_loss(lr->d, c.d_z + cfg.z_dim, cfg.z_dim + 1, L + L_CLS_D, nullptr, B, cfg.num_classes, 0, c.scratch, c.stream);
    T lf = classifier_forward(c, image);
    launch_acgan_loss(lf->d, c.d_z + cfg.z_dim, cfg.z_dim + 1, L + L_CLS_G, nullptr, B, cfg.num_classes, 1, c.scratch, c.stream);
  }
  T zr = encoder_forward(c, edge, eps_dev);
  launch_l1(c.d_z, cfg.multiclasses() ? cfg.z_dim + 1 : cfg.z_dim, zr->d, L + L_ZL, nullptr, B, cfg.z_dim, cfg.zl_weight,
            c.stream);
  c.tape.clear();
  r->edge = edge->d;
  r->image = image->d;
  r->resized[0] = r_pic ? r_pic->d : nullptr;
  r->resized[1] = r_img ? r_img->d : nullptr;
  r->resized[2] = r_edges ? r_edges->d : nullptr;
  r->resized[3] = r_edge_out ? r_edge_out->d : nullptr;
}

// edgegan.py:492-517,606-609: E(left half) -> z (+ one-hot) -> G1 ; second run -> G2 (own eps draw).
// The reference's two sess.run calls evaluate the encoder twice on the same input; mu and log-sigma are a deterministic
// function of that input, so the trunk is evaluated ONCE and only z = mu + eps * exp(log sigma) (encoder.py:78-79) is
// formed per run with that run's eps -- bit-identical outputs, 1.575 of the 3.8 GFLOP per image not executed twice.
// The two generator chains (launches of 4x4 .. 64x64 pixels, neither fills the chip) run side by side on the call's two
// lanes (main_lane, side_lane); side_lane < 0: the whole call is ONE chain on main_lane -- what eg_infer_run uses when it
// keeps four calls in flight (capi.cpp).  Everything a call touches is its lanes' own (arena, scratch, events) or read-only
// (the resident inputs, the weights and their persistent derived copies), so calls on disjoint lanes do not interact.
void infer(Ctx& c, const float* d_images, const int* d_classes, int NB, const float* d_eps2, float* d_edge,
           float* d_image, int main_lane, int side_lane) {
  const Config& cfg = c.cfg;
  comm_wait_all(c);
  c.bn_times_block = 0;   // (the test graph's moving-average updates are never saved: edgegan.py:492-517)
  const bool two = c.lanes_on && !c.prof_on && side_lane >= 0 && side_lane != main_lane;
  if (c.prof_on || !c.lanes_on) main_lane = 0;
  ensure_lane(c, main_lane);
  use_lane(c, main_lane);
  reset_step(c);
  for (int g = 0; g < G_COUNT; ++g) set_want_grad(c, g, false);
  T images = wrap_tensor(c, const_cast<float*>(d_images), NB, cfg.H, cfg.W, cfg.c_dim);
  T left = slice_w(c, images, 0, cfg.W / 2);
  T mu = nullptr, ls = nullptr;
  static const bool sc_lane = dbg_env("EG_NO_INFER_SC_LANE") == nullptr;
  if (two && sc_lane) {   // the side lane hosts the residual shortcuts of the trunk, then G2's chain (its arena: one reset per call)
    use_lane(c, side_lane);
    c.arena.reset();      // (the side lane's previous work is fully enqueued; stream order protects its memory)
    use_lane(c, main_lane);
    c.infer_side = side_lane;
  }
  T z1f = nullptr;
  T z0 = encoder_forward(c, left, d_eps2, &mu, &ls, &z1f);
  c.infer_side = -1;
  if (two) {
    warm_transposes(c, G_G2);   // G2's lazily built transposed filters: filled here, on the main lane, before the side lane forks off and reads them
    EG_HIP(hipEventRecord(c.ev_infer[main_lane], c.stream));
  }
  for (int run = 0; run < 2; ++run) {
    if (two && run == 1) {
      use_lane(c, side_lane);
      if (!sc_lane) c.arena.reset();   // (the side lane's previous work is fully enqueued; stream order protects its memory)
      EG_HIP(hipStreamWaitEvent(c.stream, c.ev_infer[main_lane], 0));
    }
    T z = run == 0 ? z0 : (z1f ? z1f : reparam(c, mu, ls, d_eps2 + 1));
    T zin = z;
    if (cfg.multiclasses()) {
      zin = new_tensor(c, NB, 1, 1, cfg.g_in(), false);
      launch_class_onehot(z->d, d_classes, zin->d, NB, cfg.z_dim, cfg.num_classes, c.stream);
    }
    generator_forward(c, run == 0 ? G_G1 : G_G2, zin, 0, nullptr, run == 0 ? d_edge : d_image);
  }
  if (two) {   // the main lane continues after the side lane
    EG_HIP(hipEventRecord(c.lanes[side_lane].done, c.stream));
    use_lane(c, main_lane);
    EG_HIP(hipStreamWaitEvent(c.stream, c.lanes[side_lane].done, 0));
  }
  use_lane(c, 0);
}

}  // namespace eg
