/* edgegan_hip.h -- C ABI of libedgegan_hip.so, the MI355X (gfx950) engine for the EdgeGAN
 * training / inference hot path.
 *
 * The reference (sysu-imsl/EdgeGAN) has no FFI / plugin interface: its device boundary is
 * tf.Session.run on a TF-1.14 graph.  Each entry point below names the reference call site it
 * replaces (paths relative to the reference repository root).  A Python maintainer binds this with
 * ctypes (INTEGRATION.md shows the stub); nothing here exposes torch / HIP types.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; eg_last_error() gives the message;
 *     no exception ever crosses this boundary;
 *   - an eg_ctx is bound to ONE HIP device and ONE stream; it is not thread-safe; use one host
 *     process (or thread) per GPU;
 *   - host buffers are C-contiguous float32 / int32 and stay owned by the caller; the library
 *     copies; device memory is owned by the library;
 *   - calls that return host data synchronise the stream; eg_train_step / eg_substep only enqueue.
 *   - image tensors are NHWC float32 in [-1,1] (edgegan/models/edgegan.py:185-186);
 *     z is [B, z_dim] or, with classes, [B, z_dim+1] whose last column is the class id
 *     (edgegan/models/edgegan.py:188-197, edgegan/utils/data/dataset.py:72-84).
 */
#ifndef EDGEGAN_HIP_H
#define EDGEGAN_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct eg_ctx eg_ctx;

/* Flags of edgegan/train.py:14-74 that shape the graph.
 * struct_size makes the layout self-describing: eg_default_config() sets it to sizeof(eg_config) of the library's build,
 * and eg_create / eg_variable_count / eg_variable_info reject any other value (code 3, "eg_config layout mismatch") -- a
 * binding compiled against an older header fails loudly instead of reading shifted fields. */
typedef struct eg_config {
  unsigned int struct_size; /* sizeof(eg_config); written by eg_default_config */
  int batch_size;      /* train.py:20  (per device)                     */
  int height;          /* train.py:21-26 input/output_height            */
  int width;           /* train.py:23-28 input/output_width (edge|image)*/
  int num_classes;     /* train.py:44-45; 0 == --nomulticlasses         */
  int z_dim;           /* train.py:74                                   */
  int gf_dim, df_dim;  /* models/edgegan.py:47                          */
  int c_dim;           /* models/edgegan.py:48                          */
  int use_image_discriminator; /* train.py:61-62; 0: D_patch2 is not built, d_optim_patch2 is not registered and
                                  image_dis_dloss = image_dis_gloss = 0 (models/edgegan.py:164-168,113-114,253-263,288-301) */
  int image_dis_size;  /* train.py:63-64                                */
  int use_edge_discriminator;  /* train.py:65-66; the same for D_patch3 / d_optim_patch3 (models/edgegan.py:170-174,264-274,303-311) */
  int edge_dis_size;   /* train.py:67                                   */
  float lambda_gp;     /* train.py:52                                   */
  float zl_weight;     /* train.py:41 stage1_zl_loss                    */
  float learning_rate; /* train.py:18                                   */
  float joint_dweight, image_dweight, edge_dweight; /* train.py:68-73   */
  int device;          /* train.py:14 --gpu                             */
  int compute_dtype;   /* no reference flag.  0 = fp32 everywhere (the reference's arithmetic), dense contractions on
                          the exact-fp32 MFMA instruction; 2 = the same fp32 arithmetic with every fp32 operand of a dense
                          conv / deconv / filter-gradient contraction split EXACTLY into three bf16 limbs and the
                          product formed from 6 exact limb products on the bf16 MFMA, fp32 accumulation (an fp32
                          implementation: held to the fp32 tolerances, ~1.35x faster -- DESIGN.md 3.4); 1 = the
                          contractions take bf16-ROUNDED operands on the bf16 MFMA with fp32 accumulation; storage,
                          norms, losses and RMSProp stay fp32 in every mode (BASELINE configs 2,4,5) */
  int e_norm, g_norm, d_norm; /* train.py:54-59 E_norm / G_norm / D_norm: 0 = 'instance' (the default), 1 = 'batch', 2 = None
                                 (normalization.py:26-27: the block keeps its activation only; reachable from Python, not from
                                 the string flags, whose 'None' fails normalization.py:12's assertion) -- the
                                 normalisation of the encoder's residual blocks (models/encoder.py:67), the generators'
                                 transposed-convolution blocks (models/generator.py:54-68; the first norm is BatchNorm whatever
                                 the flag, generator.py:51-52) and the discriminators' blocks 2 - 4 (models/discriminator.py:
                                 63-72), nn/modules/normalization.py:10-29.  'batch' adds the contrib batch_norm variables
                                 <block>/batch_norm/BatchNorm/{beta, gamma, moving_mean, moving_variance}. */
} eg_config;

int eg_default_config(eg_config* cfg);

/* ---- variable registry: replaces tf.get_variable creation + tf.train.Saver's view of the graph
 *      (models/edgegan.py:421; names per SURVEY.md Appendix C).  Usable without a GPU. ---------- */
int eg_variable_count(const eg_config* cfg);
/* init_kind: 0 constant(init_param) 1 normal(0,init_param) 2 truncated normal(0,init_param)
 *            3 xavier/glorot uniform.  shape has up to 4 dims. */
int eg_variable_info(const eg_config* cfg, int index, char* name, int name_cap, int* shape, int* ndim,
                     int* trainable, int* init_kind, float* init_param);

/* ---- lifecycle: replaces tf.Session + EdgeGAN.build_train_model (models/edgegan.py:413-423) --- */
int eg_create(const eg_config* cfg, eg_ctx** out);
int eg_destroy(eg_ctx* ctx);
const char* eg_last_error(eg_ctx* ctx); /* ctx may be NULL: last error of a failed eg_create */

/* ---- variables: replaces saver.restore / sess.run(var) (models/edgegan.py:635-657).
 *      `name` is the TF variable name; suffix "/RMSProp" = rms slot, "/RMSProp_1" = momentum slot,
 *      "/grad" (get only) = gradient left by the last eg_substep(..., apply=0). ---------------- */
int eg_set_variable(eg_ctx* ctx, const char* name, const float* host, long long numel);
int eg_get_variable(eg_ctx* ctx, const char* name, float* host, long long numel);

/* ---- training: replaces EdgeGAN.update_model (models/edgegan.py:126-130) ----------------------
 * eg_upload_batch makes one feed resident in HBM (the feed_dict of models/edgegan.py:130):
 *   images [B,H,W,3]; z [B,z_dim(+1)]; alpha [3,B] ~ U(0,1), the random_blend draws of the joint /
 *   image / edge discriminator runs (models/edgegan.py:32-35); eps [1] ~ N(0,1), the encoder's
 *   scalar noise of the e_optim run (models/encoder.py:78-79).
 * eg_substep runs optimizer sub-step k = 1..7 in the registration order of
 *   models/edgegan.py:109-124: d_optim, d_optim_patch2, d_optim_patch3, d_optim2, g_optim_u,
 *   e_optim, g_optim_b; apply=0 leaves the weights untouched and the gradients readable.
 * eg_train_step = sub-steps 1..7 with apply=1 (one iteration of models/edgegan.py:461). */
int eg_upload_batch(eg_ctx* ctx, const float* images, const float* z, const float* alpha, const float* eps);
int eg_substep(eg_ctx* ctx, int k, int apply);
int eg_train_step(eg_ctx* ctx);
/* losses of the last sub-steps, replaces the evaluate() read-backs of models/edgegan.py:464-480:
 * out[0..11] = joint wgan, joint gp, image wgan, image gp, edge wgan, edge gp, loss_d_ac,
 *              joint_dis_gloss, edge_dis_gloss, image_dis_gloss, loss_g_ac, zl_loss */
int eg_read_losses(eg_ctx* ctx, float* out12);
/* The loop of models/edgegan.py:456-486 (feed a batch, run the sub-steps, read the losses back) without a host<->device
 * synchronisation per iteration: eg_train_batch = eg_upload_batch + eg_train_step with the feed staged through pinned
 * host memory and enqueued behind the previous iteration (returns at once; the host buffers may be reused on return);
 * eg_losses_wait(back, out12) returns the losses of the latest (back = 0) or the previous (back = 1) eg_train_batch,
 * blocking only until that iteration has finished.  A loop that enqueues iteration k and then reports iteration k-1
 * keeps the GPU busy while the host decodes, prints and writes summaries. */
int eg_train_batch(eg_ctx* ctx, const float* images, const float* z, const float* alpha, const float* eps);
int eg_losses_wait(eg_ctx* ctx, int back, float* out12);

/* Post-update evaluation with the reference's semantics -- replaces the observe-only runs the reference issues AFTER
 * update_model: sess.run(d_sum), sess.run(g_sum) (models/edgegan.py:427-433) and the five evaluate() calls
 * (models/edgegan.py:464-480).  Forward-only on the resident batch with the CURRENT weights; alpha [3][B] and eps are
 * the fresh random_blend / encoder-noise draws those runs make (caller-supplied); bn_updates = how many times the
 * generators' BatchNorm moving statistics are advanced (contrib batch_norm updates them in every run that evaluates
 * a generator: 6 mirrors one iteration of the reference loop, 0 leaves them alone).  losses12 as eg_read_losses;
 * every other pointer is optional (NULL = not wanted) and feeds the histogram / image summaries of
 * models/edgegan.py:344-411: d_logits [6][B] = joint real, joint fake, image real, image fake, edge real, edge fake
 * (D_logits; "d" / "d_" are their sigmoids); edge_output / image_output [B,H,W/2,3] ("G1", "G2"); the four resized
 * tensors [B,dis,dis,3] ("resized_inputs_image", "resized_G_image", "resized_inputs_p3_image", "resized_G_p3_image"). */
typedef struct eg_eval_outputs {
  float* losses12;
  float* d_logits;
  float* edge_output;
  float* image_output;
  float* resized_inputs;
  float* resized_image_output;
  float* resized_edges;
  float* resized_edge_output;
} eg_eval_outputs;
int eg_eval_losses(eg_ctx* ctx, const float* alpha, float eps, int bn_updates, const eg_eval_outputs* out);

/* ---- inference: replaces the two sess.run calls of EdgeGAN.test (models/edgegan.py:606-609) on
 *      the graph of define_test_input (models/edgegan.py:492-517).  images [nb,H,W,3], classes [nb]
 *      (ignored without classes), eps2 = the two scalar encoder draws; outputs [nb,H,W/2,3]. ---- */
int eg_infer(eg_ctx* ctx, const float* images, const int* classes, int nb, const float* eps2, float* edge_out,
             float* image_out);
/* the same in three steps, for callers that keep a batch resident in HBM (serving loops, bench.py --mode infer):
 * upload = the feed_dict copy, run = the two sess.run calls (enqueue only), download = the fetch + sync.
 * eg_infer is exactly upload + run + download. */
int eg_infer_upload(eg_ctx* ctx, const float* images, const int* classes, int nb, const float* eps2);
int eg_infer_run(eg_ctx* ctx);
int eg_infer_download(eg_ctx* ctx, float* edge_out, float* image_out);

int eg_sync(eg_ctx* ctx);

/* ---- data parallel (no reference counterpart; the reference is single-GPU, train.py:77) -------
 * rank 0 calls eg_comm_unique_id and ships the 128 bytes to the other ranks by any host channel;
 * every rank then calls eg_comm_init.  eg_train_step then all-reduces (RCCL, sum) each optimised
 * group's flat gradient buffer and averages before RMSProp. */
int eg_comm_unique_id(eg_ctx* ctx, char* out128);
int eg_comm_init(eg_ctx* ctx, const char* id128, int rank, int nranks);
/* all-reduce of n <= 64 host floats over the ranks (op: 0 sum, 2 max, 3 min); returns when every rank has arrived,
 * i.e. it is also the host barrier of the data-parallel loop -- the host needs no other collective library */
int eg_comm_reduce_scalars(eg_ctx* ctx, float* inout, int n, int op);
/* introspection of the exchange (no reference counterpart): ranks, RCCL version code (ncclGetVersion, 0 before
 * eg_comm_unique_id / eg_comm_init), the path the bound librccl was mapped from, and the gradient all-reduces enqueued so
 * far with their payload in bytes -- bench.py prints them so that a multi-GPU line is self-describing */
int eg_comm_info(eg_ctx* ctx, int* nranks, int* rccl_version, char* libpath, int libpath_cap, long long* allreduce_calls,
                 long long* allreduce_bytes);

/* per-collective timing of the gradient exchange (no reference counterpart).  eg_comm_timing(ctx, 1) brackets every gradient
 * all-reduce enqueued from now on with a HIP event pair on the communication stream; eg_comm_timing_read waits for that stream and
 * returns, in issue order, for up to `cap` collectives since the last read: the network group (0 G1+G2 joint exchange, 2 D2, 3 D,
 * 4 D_patch2, 5 D_patch3, 6 E), the payload in bytes and the milliseconds between the stream reaching the collective and its
 * result being complete (the wait for the slowest rank is inside) -- bench.py prints the per-group mean, max over the ranks.
 * Environment (read at eg_create / eg_comm_init, documented in csrc/comm.cpp): EG_COMM_CHANNELS=n caps RCCL's channels (=
 * workgroups = compute units taken from the matrix kernels), EG_COMM_CUS=n confines the communication stream to n compute units by
 * a CU mask; RCCL's own NCCL_* / RCCL_* variables pass through. */
int eg_comm_timing(eg_ctx* ctx, int enable);
int eg_comm_timing_read(eg_ctx* ctx, float* ms, int* group, long long* bytes, int cap, int* n);

/* ---- measurement -------------------------------------------------------------------------------
 * Between begin/end every GEMM-shaped launch is bracketed by HIP events on the engine's stream;
 * end writes a JSON object {"kernels":[{"name","launches","ms","flops"}...]} into buf. */
int eg_profile_begin(eg_ctx* ctx);
int eg_profile_end(eg_ctx* ctx, char* buf, int cap);
/* one CSV line per recorded launch: "<variable> <fwd|dgrad|wgrad>", kernel, GEMM shape, microseconds, GFLOP --
 * the per-layer roofline table of profiles/ (tools/layer_table.py) */
int eg_profile_layers(eg_ctx* ctx, char* buf, long long cap);
int eg_memory_stats(eg_ctx* ctx, long long* arena_peak_bytes, long long* arena_reserved_bytes);
/* TFLOP/s the matrix pipe of this device sustains on a memory-free v_mfma_f32_32x32x16_bf16 loop (one 8-wave block per
 * CU, iters x 16 MFMAs per wave), with constant (random_operands = 0) or pseudo-random bf16 operands: the data toggling
 * sets the pipe's power and with it its clock.  bench.py prints both next to roofline.peak (no reference counterpart). */
int eg_mfma_probe(eg_ctx* ctx, int iters, int random_operands, double* tflops);

/* ---- per-op / per-network test entry points (parity tests against oracle/) ----------------------
 * eg_test_op runs ONE tape op forward and, if a cotangent is supplied, backward, through exactly
 * the code the training step uses.  See tests/test_ops_gpu.py for the per-op argument lists.
 * Composite ops (round 5) build the op SEQUENCES of the classifier's MRU unit / the discriminator head the way
 * classifier_forward / disc_forward do, so that the fused forms are exercised as the graph uses them: "mru_tail"
 * (conv.py:217-243: Conv_1 + PReLU, Conv_2, pool2_add + the next unit's PReLU; a sixth integer parameter of 1 freezes its
 * parameters as the generator sub-steps freeze the classifier's: input gradients only), "mru_gate" (conv.py:185-209: prelu(h),
 * the update gate as image rows + hidden rows, min-max, gate + PReLU, a convolution behind it), "act_conv" (InstanceNorm
 * + LeakyReLU or pool2 in front of a convolution), "d_head" (discriminator.py:61-66: d_conv_0 + LeakyReLU, d_conv_1);
 * argument lists in csrc/capi.cpp, oracle restatements in tests/parity.py.
 * eg_net_forward runs one network forward: net in {"G1","G2","D","D_patch2","D_patch3","E","D2"}. */
int eg_test_op(eg_ctx* ctx, const char* op, int n_in, const float* const* inputs, const long long* in_numel,
               int n_iparam, const int* iparams, int n_out, float* const* outputs, const long long* out_numel);
int eg_net_forward(eg_ctx* ctx, const char* net, const float* input, int nb, int h, int w, int ch, float eps,
                   float* out, long long out_numel);

/* ---- checkpoint IO helper: CRC-32C (Castagnoli) as used by TensorFlow's tensor-bundle files that
 *      tf.train.Saver writes (models/edgegan.py:635-639); needs no GPU. -------------------------------- */
unsigned int eg_crc32c(unsigned int crc, const void* data, unsigned long long n);

#ifdef __cplusplus
}
#endif
#endif /* EDGEGAN_HIP_H */
