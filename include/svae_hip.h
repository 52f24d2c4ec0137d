/* svae_hip.h -- C-ABI of libsvae_hip.so: the MI355X (gfx950) kernels of the SVAE training / scoring step.
 *
 * Reference: noveens/svae_cf, main_svae.ipynb ("nb:N" = raw line N of that file).  The reference has
 * no FFI of its own: its hot path is the PyTorch call sites listed in SURVEY.md section 2a.  Each
 * entry point below names the reference call site(s) it replaces; INTEGRATION.md shows the ctypes
 * binding a maintainer of the notebook would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless stated otherwise;
 *   - every function enqueues on the caller's hipStream_t (passed as void*), never allocates, never
 *     frees and never synchronises: workspaces are arguments;
 *   - returns 0 on success, < 0 for an argument error, > 0 = hipError_t of a failed launch;
 *     svae_last_error() returns the message of the calling thread's last failure;
 *   - fp32 row-major tensors; model dimensions are compile-time (hyper_params nb:82-85):
 *       E = 256 item_embed_size, H = 200 rnn_size (padded HP = 224), D = 150 hidden_size (padded
 *       DP = 160), L = 64 latent_size; gate columns G = 3*224 = 672 ordered (r, z, n) with gate g of
 *       hidden unit j at column g*224 + j.  Pad rows / columns of weights are zero;
 *   - a batch is TIME-MAJOR PACKED: users sorted by length (descending), nact[t] users are active at
 *     timestep t, row_off[t] = first row of timestep t, R = row_off[Tmax] rows in total;
 *   - prec: 0 = exact fp32 MFMA (v_mfma_f32_*_f32), 1 = bf16 MFMA with fp32 accumulation.
 */
#ifndef SVAE_HIP_H
#define SVAE_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- plumbing ------------------------------------------------------------------------------ */
const char* svae_last_error(void);
int svae_abi_version(void);
int svae_check_device(void); /* 0 if the current device is gfx950 */

/* ---- generic GEMM: C[M,N] = epi(sum_k A(m,k) B(k,n))  -------------------------------------------
 * Replaces the nn.Linear call sites (addmm): Encoder.linear1 nb:694, Model.linear1 nb:738,
 * Decoder.linear1 nb:708, Decoder.linear2 nb:710 (dense logits for the Model.forward surface), the
 * GRU input GEMM inside nn.GRU nb:759, and their autograd GEMMs (nb:1205).
 * a_layout 0: A stored [M][K]; 1: A stored [K][M].  b_layout 0: B stored [N][K]; 1: B stored [K][N].
 * act 0 none, 1 tanh, 2 multiply by (1 - aux^2) (tanh backward).  nsplit > 1: split-K, split z writes
 * C + z*slab_stride (no epilogue); reduce with svae_reduce_slabs.  colsum (optional, a_layout 1):
 * colsum[m] = sum_k A(m,k) -- the bias gradient of a weight-gradient GEMM, folded into the same pass. */
int svae_gemm(const float* A, const float* B, float* C, const float* bias, const float* aux, float* colsum, int M,
              int N, int K, int lda, int ldb, int ldc, int ldaux, int a_layout, int b_layout, int act, int prec,
              int nsplit, long long slab_stride, void* stream);
int svae_reduce_slabs(const float* slabs, float* out, long long n, int nsplit, long long slab_stride, void* stream);
/* Decoder.linear2 as Model.forward returns it (nb:710, nb:752-768): dense logits of n rows of the decoder hidden,
 * logits[r][i] = h2[r] . W3[i] + b3[i]  (h2 [n][160], W3 [I][160], logits [n][ldl]).  The notebook surface and
 * evaluate() rank these; training never materialises them.  prec 0: exact fp32, 1: bf16 operands. */
int svae_logits_fwd(const float* h2, const float* w3, const float* b3, float* logits, int n, int I, int ldl, int prec,
                    void* stream);
/* Weight + bias gradient of a Linear layer (autograd of y = x W^T + b, nb:1205), bf16 MFMA path:
 * gW[n][k] = sum_r dy[r][n] x[r][k], gb[n] = sum_r dy[r][n].  The rows are split nsplit ways; split z writes
 * gW at out + z*slab_stride and gb at colsum + z*slab_stride (colsum may be NULL); sum with svae_reduce_slabs.
 * See csrc/wgrad.hip. */
int svae_wgrad(const float* dy, const float* x, float* out, float* colsum, int R, int N, int K, int ldy, int ldx,
               int nsplit, long long slab_stride, void* stream);
/* dy's logical column n is stored column n + (n >= remap_from ? remap_add : 0) */
int svae_wgrad_cols(const float* dy, const float* x, float* out, float* colsum, int R, int N, int K, int ldy, int ldx,
                    int nsplit, long long slab_stride, int remap_from, int remap_add, void* stream);

/* Row-panel form of the two GRU-side contractions (bf16 MFMA path; csrc/rowgemm.hip), replacing the same call
 * sites as svae_gemm does for them: which 0: gi [R][672] = emb [R][256] W_ih^T + b_ih (input GEMM of nn.GRU,
 * nb:759); which 1: demb [R][256] = dgi [R][672] W_ih (its data gradient, nb:1205).  svae_rowgemm_prep lays
 * W_ih ([672][256] fp32 as stored) out as bf16 fragments (svae_rowgemm_frag_elems(which) elements) whenever it
 * has changed. */
int svae_rowgemm_frag_elems(int which);
int svae_rowgemm_prep(const float* wih, int which, void* frags, void* stream);
int svae_rowgemm(const float* x, const void* frags, const float* bias, float* y, int R, int which, void* stream);
int svae_rowgemm_ld(const float* x, int ldx, const void* frags, const float* bias, float* y, int R, int which,
                    void* stream);   /* rows of x ldx floats apart */

int svae_rowgemm_bf16x(const void* x, int ldx, const void* frags, float* y, int R, void* stream);   /* which 1, bf16 x */
int svae_wgrad_cols_bf16dy(const void* dy, const float* x, float* out, float* colsum, int R, int N, int K, int ldy,
                           int ldx, int nsplit, long long slab_stride, int remap_from, int remap_add, void* stream);

/* ---- item embedding ----------------------------------------------------------------------------
 * svae_embed_fwd replaces self.item_embed(x) nb:756: out[r][:] = table[x[r]][:] (E = 256).
 * svae_embed_bwd replaces the embedding backward of loss.backward() nb:1205 as a deterministic
 * segmented sum: perm = rows sorted by item, segment s covers perm[seg_off[s]..seg_off[s+1]) and is
 * written to gtable[seg_item[s]] (gtable zero-filled by the caller).  Segments are ordered by length,
 * longest first; the first nlong of them have more than svae_embed_long_rows() rows (a workgroup each). */
int svae_embed_fwd(const float* table, const int* x, float* out, int R, int I, void* stream);
int svae_embed_long_rows(void);
int svae_embed_bwd(const float* demb, const int* perm, const int* seg_off, const int* seg_item, float* gtable,
                   int nseg, int nlong, float scale, void* stream);

/* ---- GRU recurrence, persistent over timesteps ---------------------------------------------------
 * svae_gru_fwd replaces the recurrent part of nn.GRU (call nb:759, definition nb:724-727; h0 = 0).
 * gi [R][672] = emb W_ih^T + b_ih (svae_gemm); whh [672][224], bhh [672].
 * Outputs hs [R][224], hprev [R][224] (h_{t-1}), gates4 [R][224][4] = (r, z, n, gh_n).
 * svae_gru_bwd replaces autograd through nn.GRU (nb:1205): whh_t = W_hh^T [224][672];
 * dhs [R][224] in; dgi [R][672] = d/d gi, dgh [R][672] = d/d gh out. */
int svae_gru_fwd(const float* gi, const float* whh, const float* bhh, const int* row_off, const int* nact, int B,
                 int Tmax, float* hs, float* hprev, float* gates4, int prec, void* stream);
/* Timesteps [t0, t1) of the same recurrence; t0 > 0 continues from hs (bf16 kernel only).  No counterpart in the
 * reference (one nn.GRU call, nb:759): the host runs the second range beside the decoder side of the first.
 * prec 1 / 2: hprev and gates4 may BOTH be NULL when no backward follows (scoring): the saved state is not written. */
int svae_gru_fwd_range(const float* gi, const float* whh, const float* bhh, const int* row_off, const int* nact, int B,
                       int Tmax, int t0, int t1, float* hs, float* hprev, float* gates4, int prec, void* stream);
/* svae_gru_bwd with dgh == NULL (bf16 kernel): dgi is ONE array [R][896] = (dr, dz, dn, dn*r) that carries both
 * gradients (they only differ in the third gate); consumers: svae_rowgemm_ld / svae_wgrad_cols. */
/* the same with bf16 elements (every consumer rounds to bf16 anyway): svae_rowgemm_bf16x / svae_wgrad_cols_bf16dy */
int svae_gru_bwd_bf16out(const float* dhs, const float* hprev, const float* gates4, const float* whh_t,
                         const int* row_off, const int* nact, int B, int Tmax, void* dg4, void* stream);
/* The BPTT over timesteps [t0, t1) only (it runs from t1-1 down to t0; bf16 kernel, merged bf16 gradient array).
 * t1 < Tmax continues a BPTT whose steps >= t1 have been run with the same dh_carry [B][224] fp32; t0 > 0 leaves the
 * recurrent part of dL/dh_{t0-1} there.  When the call has run, the rows of timesteps >= t0 of dg4 are final.  No
 * counterpart in the reference (one autograd call, nb:1205): the split lets the weight and embedding gradients of the
 * late timesteps' rows start while the BPTT of the early ones is still running. */
int svae_gru_bwd_range_bf16out(const float* dhs, const float* hprev, const float* gates4, const float* whh_t,
                               const int* row_off, const int* nact, int B, int Tmax, int t0, int t1, void* dg4,
                               float* dh_carry, void* stream);
int svae_gru_bwd(const float* dhs, const float* hprev, const float* gates4, const float* whh_t, const int* row_off,
                 const int* nact, int B, int Tmax, float* dgi, float* dgh, int prec, void* stream);

/* ---- latent: reparameterisation + KL ---------------------------------------------------------------
 * Replaces Model.sample_latent nb:734-750 and the KL term of VAELoss.forward nb:797.
 * tmp [R][128] = (mu | log_sigma); z = mu + exp(log_sigma) * eps; kl_row[r] = sum_64 0.5(-ls+exp(ls)+mu^2-1).
 * Backward: dtmp = (dz + a*w_kl[r]*mu | dz*exp(ls)*eps + a*w_kl[r]*0.5*(exp(ls)-1)), a = anneal. */
int svae_latent_fwd(const float* tmp, const float* eps, float* z, float* kl_row, int R, void* stream);
int svae_latent_bwd(const float* tmp, const float* eps, const float* dz, const float* w_kl, float anneal,
                    float* dtmp, int R, void* stream);
/* Replaces np.random.normal(0,1,size) nb:744 by Philox4x32-10 + Box-Muller keyed by (seed, step, row_key[r], col). */
int svae_philox_normal(float* out, const long long* row_key, int R, unsigned long long seed, unsigned step,
                       void* stream);

/* ---- fused row-wise MLP chain (bf16 MFMA path) --------------------------------------------------------
 * svae_mlp_fwd replaces Encoder.forward nb:693-696, Model.sample_latent nb:734-750, the KL row term of
 * VAELoss nb:797 and Decoder.linear1 + tanh nb:708-709 in one kernel: hs [R][224] -> enc [R][160],
 * tmp [R][128] = (mu | log_sigma), z [R][64], d1 [R][160], kl_row [R].  svae_mlp_bwd replaces their autograd
 * (data gradients, nb:1205): dd1 [R][160] -> dtmp [R][128], denc [R][160], dhs [R][224].
 * svae_mlp_prep lays the three weights out as bf16 MFMA fragments (svae_mlp_frag_elems() elements); call it
 * whenever the weights changed.  See csrc/mlp.hip. */
int svae_mlp_frag_elems(void);
int svae_mlp_prep(const float* w1, const float* wl, const float* w2, void* frags, void* stream);
int svae_mlp_fwd(const float* hs, const float* eps, const void* frags, const float* b1, const float* bl,
                 const float* b2, float* enc, float* tmp, float* z, float* d1, float* kl_row, int R, void* stream);

/* svae_mlp_fwd over rows [row0, row1) of the batch (same reference lines) that also writes the projection sweeps'
 * bf16 operand copies of d1 -- what svae_proj_prep_rows_range makes from d1, bit for bit -- so the decoder side
 * runs one kernel less.  Pointers are array bases; row0 % 32 == 0; row1 <= R_pad (R_pad % 64 == 0).
 * enc, tmp and z may all be NULL when no backward follows (scoring): they are then not written. */
int svae_mlp_fwd_prep(const float* hs, const float* eps, const void* frags, const float* b1, const float* bl,
                      const float* b2, float* enc, float* tmp, float* z, float* d1, float* kl_row, int R, int R_pad,
                      int row0, int row1, void* h2b, void* h2s, void* h2tb, void* stream);
int svae_mlp_bwd(const float* dd1, const float* tmp, const float* eps, const float* enc, const float* w_kl,
                 const void* frags, float anneal, float* dtmp, float* denc, float* dhs, int R, void* stream);

/* ---- loss over materialised logits (fp32 path) ------------------------------------------------------
 * Replaces F.log_softmax nb:802 + likelihood nb:805-808 (+ their gradient when write_grad != 0:
 * logits are overwritten by w_nll[r] * (n_r softmax - y)).  tgt [R][K] int32, -1 = padding. */
int svae_softmax_nll(float* logits, const int* tgt, const float* w_nll, float* nll_row, float* lse, int R, int I,
                     int ldI, int K, int write_grad, void* stream);
/* out3 = {sum_r w_nll nll_row + anneal * sum_r w_kl kl_row, likelihood part, KL part}  (VAELoss nb:810) */
int svae_loss_reduce(const float* nll_row, const float* kl_row, const float* w_nll, const float* w_kl, float anneal,
                     int R, float* out3, void* stream);

/* ---- fused projection + log-softmax + NLL (bf16 MFMA path; logits are never materialised) -----------
 * Replace Decoder.linear2 nb:710 + F.log_softmax nb:802 + likelihood nb:805-808 and their autograd
 * (loss.backward() nb:1205).  See csrc/proj.hip for the operand copies and their constant columns.
 *   svae_proj_prep_rows   bf16 copies of the decoder hidden rows h2 [R][160] (tanh of Decoder.linear1 nb:709):
 *                         h2b plain, h2s scaled by log2(e) (both [R_pad][160]), h2tb transposed [160][R_pad]
 *   svae_proj_prep_items  the same for Decoder.linear2's weight w3 [I][160] and bias b3: w3b, w3s, w3tb
 *   svae_proj_nll_fwd     loss only (scoring): nll_row[R], lse[R];  part = 32 R floats of scratch
 *   svae_proj_nll_fwd_dh  forward AND the hidden-side gradient in one sweep over the items: nll_row, lse,
 *                         dd1 [R][160] = dL/d(pre-tanh decoder hidden); writes the row constants into h2b.
 *                         slabs = 16 R 160 floats, mpart = 16 R floats of scratch
 *   svae_proj_nll_bwd_dw  weight-side gradient gw3 [I_rows][160], gb3 [I_rows] (after _fwd_dh of the same
 *                         batch); slabs = 16 I 160 floats; occ_off [I+1] / occ_row = target occurrences by item,
 *                         occ_items [I] = items by occurrence count (descending), the first nocc_long of them
 *                         with more than svae_proj_occ_long() occurrences
 * The transposed copies h2tb / w3tb are read by the split instantiation (prec 2) only -- the bf16 sweeps take Y^T out
 * of the row-major image with transposed LDS reads -- and may be NULL everywhere when the part strides are 0.
 * R_pad, I_pad: multiples of 64.  parts (bit mask, 3 = everything): 1 sweep kernel, 2 finish kernel -- the
 * split exists so a profiler can bracket one kernel.
 * Work division: every sweep takes a schedule built ONCE per shape by svae_proj_schedule (a HOST function:
 * nx resident rows -- R for _fwd / _fwd_dh, I for _bwd_dw -- ny streamed rows, ncu compute units; uniform = 0:
 * balanced form, nominal 8-way split; uniform = -n: balanced form with a nominal n-way split; uniform = 1
 * gives every x tile the same split count (for sweeps whose streamed operand outgrows the L2); writes 4 ints
 * per workgroup into HOST memory `sched`, room for cap >= svae_proj_ns_max() * ceil(nx/256) workgroups; returns
 * the workgroup count nwg, and ns_q / ns_rem: x tile j of 256 rows is split ns_q + (j < ns_rem) ways).  The
 * caller copies the table to the device and passes it with (nwg, ns_q, ns_rem).  Scratch is sized for
 * svae_proj_ns_max() (= 16) slabs: part 16*2*R, mpart 16*R, slabs 16*R*160 (_fwd_dh) or 16*I*160 (_bwd_dw) floats. */
int svae_proj_nsplit(void);
int svae_proj_ns_max(void);
int svae_proj_schedule(int nx, int ny, int ncu, int uniform, int* sched_host, int cap, int* ns_q, int* ns_rem);
/* FLAT work division (round 3; _fwd_dh and _bwd_dw in their bf16 instantiation): the (x tile, y tile) space, x-major,
 * cut into nwg_want contiguous ranges of equal cost, one per workgroup; a range that crosses x-tile boundaries is run
 * segment by segment by the same workgroup (a further segment costs `boundary` tiles' worth: 5 measured).  Writes
 * into HOST memory `sched` (room for cap4 int4): 4 ints per segment (x tile, first y tile, end y tile, slab), the
 * workgroups' segment offsets [nwg + 1] and the slab count of every x tile (each padded to a multiple of 4 ints).
 * Returns the table length in int4 units, *nwg (<= nwg_want) and *nseg; the sweeps take the device copy of the table
 * as (sched, nwg, ns_q = 0, ns_rem = nseg).  cap4 >= (nwg_want + x tiles) * 5 / 4 + 4 always suffices (x tiles = ceil(nx/256)). */
int svae_proj_schedule_flat(int nx, int ny, int nwg_want, int boundary, int* sched_host, int cap4, int* nwg, int* nseg);
int svae_proj_tile(void);
int svae_proj_prep_rows(const float* h2, int R, int R_pad, void* h2b, void* h2s, void* h2tb, void* stream);
/* rows [row0, row1) only (multiples of 32; pointers of the whole arrays) */
int svae_proj_prep_rows_range(const float* h2, int R, int R_pad, int row0, int row1, void* h2b, void* h2s, void* h2tb,
                              void* stream);
int svae_proj_prep_items(const float* w3, const float* b3, int I, int I_pad, void* w3b, void* w3s, void* w3tb,
                         void* stream);
int svae_proj_nll_fwd(const void* h2s, const void* w3b, const float* h2, const float* w3, const float* b3,
                      const int* tgt, int R, int I, int K, float* part, float* nll_row, float* lse,
                      const int* sched, int nwg, int ns_q, int ns_rem, void* stream);
int svae_proj_nll_fwd_dh(const void* h2s, void* h2b, const void* w3b, const void* w3tb, int I_pad, const float* h2,
                         const float* w3, const float* b3, const int* tgt, const float* w_nll, int R, int I, int K,
                         float* slabs, float* mpart, float* nll_row, float* lse, float* dd1, const int* sched,
                         int nwg, int ns_q, int ns_rem, int parts, void* stream);
int svae_proj_occ_long(void);
int svae_proj_nll_bwd_dw(const void* w3s, const void* h2b, const void* h2tb, int R, int R_pad, int I, int I_rows,
                         const float* h2, const float* w_nll, const int* occ_off, const int* occ_row,
                         const int* occ_items, int nocc_long, float* slabs, float* gw3, float* gb3,
                         const int* sched, int nwg, int ns_q, int ns_rem, int parts, void* stream);

/* ---- split instantiations ("prec 2") of the fused bf16 kernels ---------------------------------------------
 * Same reference call sites as the entry points they extend (nb:694-710, nb:759, nb:802-808, nb:1205).  Every bf16
 * operand is carried as hi + mid + lo parts (mid = RNE(v - hi), lo = RNE(v - hi - mid): all 24 significant bits)
 * and every product by its six leading terms hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi, so the SAME kernel
 * templates (tiling, work division, constant columns, running stabiliser, slab / finish structure) deliver
 * fp32-class results: the parity gates of tests/ run through them.  Six times the MFMA work and a simple one-tile
 * LDS pipeline in the sweeps: never timed.
 *   prec argument: 1 = bf16 (identical to the entry point without the suffix), 2 = split.
 *   part strides (projection copies): the mid and lo parts of a bf16 operand copy sit part_stride and
 *   2 part_stride ELEMENTS behind it, same shape and layout; one stride for the three row copies (h2b / h2s /
 *   h2tb), one for the three item copies (w3b / w3s / w3tb); 0 = bf16 copies only.
 *   fragment buffers (svae_rowgemm_prep_prec, svae_mlp_prep_prec) hold the hi fragments followed by the mid and
 *   the lo fragments: three times svae_rowgemm_frag_elems / svae_mlp_frag_elems elements when prec == 2. */
int svae_wgrad_cols_prec(const float* dy, const float* x, float* out, float* colsum, int R, int N, int K, int ldy,
                         int ldx, int nsplit, long long slab_stride, int remap_from, int remap_add, int prec,
                         void* stream);
int svae_rowgemm_prep_prec(const float* wih, int which, void* frags, int prec, void* stream);
/* svae_embed_fwd over rows [0, R) and svae_rowgemm_prep_prec(wih, 0, frags, 1) in ONE launch: the two independent
 * kernels that head a training step (reference: nn.Embedding call nb:756, the GRU's input weights nb:724).
 * out_bf16 != 0: the gathered rows are written as bf16 ([R][256] two-byte elements) -- what both consumers of the rows
 * round them to anyway -- for svae_rowgemm_gi_bf16x (gi = emb W_ih^T + b_ih) and svae_wgrad_cols_bf16dy_bf16x (the
 * input-weight gradient), bit-identical to the fp32-row forms.  svae_embed_fwd_bf16: the gather alone, bf16 rows. */
int svae_embed_fwd_wih_prep(const float* table, const int* x, void* out, int R, int I, const float* wih, void* frags,
                            int out_bf16, void* stream);
int svae_embed_fwd_bf16(const float* table, const int* x, void* out16, int R, int I, void* stream);
int svae_rowgemm_gi_bf16x(const void* x16, int ldx, const void* frags, const float* bias, float* y, int R, void* stream);
int svae_wgrad_cols_bf16dy_bf16x(const void* dy16, const void* x16, float* out, float* colsum, int R, int N, int K, int ldy,
                                 int ldx, int nsplit, long long slab_stride, int remap_from, int remap_add, void* stream);
/* The GRU's recurrent- and input-weight gradients (two of the products inside loss.backward(), nb:1205) in ONE launch:
 * job 0 = svae_wgrad_cols_bf16dy(dy, x0, ...), job 1 = svae_wgrad_cols_bf16dy(dy, x1, ...) or, x1_bf16 != 0,
 * svae_wgrad_cols_bf16dy_bf16x(dy, x1, ...); both share dy, R, N, ldy and nsplit and write the slabs the single
 * launches write (bit-identical).  svae_reduce_slabs_pair sums both slab sets in one launch. */
int svae_wgrad_pair_bf16dy(const void* dy, int R, int N, int ldy, int nsplit,
                           const float* x0, int K0, int ldx0, float* out0, float* colsum0, long long stride0,
                           int remap_from0, int remap_add0,
                           const void* x1, int x1_bf16, int K1, int ldx1, float* out1, float* colsum1, long long stride1,
                           int remap_from1, int remap_add1, void* stream);
/* The same two jobs with BOTH X operands as bf16 elements, on 256 x 256 output tiles (one wave per SIMD, 32 MFMAs per
 * wave and 32-row stage); x0 = the bf16 copy of the previous hidden states made by svae_f32_to_bf16.  Faster than the
 * pair launch above alone (36 against 48 us at 21.8k rows), not inside the engine's step (its workgroups need a whole
 * CU each), so the engine takes it only on request (SVAEEngine.wgrad_big).  Same slab layout; the split count may differ from the 128 x 128 kernels' (40 = 240 workgroups, one
 * round), so results agree with them to fp32 summation order, not bit for bit. */
int svae_wgrad_pair_big(const void* dy, int R, int N, int ldy, int nsplit,
                        const void* x0, int K0, int ldx0, float* out0, float* colsum0, long long stride0,
                        int remap_from0, int remap_add0,
                        const void* x1, int K1, int ldx1, float* out1, float* colsum1, long long stride1,
                        int remap_from1, int remap_add1, void* stream);
/* svae_wgrad_pair_bf16dy with BOTH X operands as bf16 elements (x0: the bf16 copy of the previous hidden states), on the
 * all-bf16 128 x 128 kernel with hand-counted waits (no drain of the LDS-DMA ring at every stage): same workgroups and
 * slabs as the pair launch; weight gradients bit-identical to svae_wgrad_cols_bf16dy_bf16x's, bias gradients equal to
 * summation order. */
int svae_wgrad_pair_bf16(const void* dy, int R, int N, int ldy, int nsplit,
                         const void* x0, int K0, int ldx0, float* out0, float* colsum0, long long stride0,
                         int remap_from0, int remap_add0,
                         const void* x1, int K1, int ldx1, float* out1, float* colsum1, long long stride1,
                         int remap_from1, int remap_add1, void* stream);
/* dst [n] (bf16) = round-to-nearest-even of src [n] (fp32); n a multiple of 8, both 16-byte aligned. */
int svae_f32_to_bf16(const float* src, void* dst, long long n, void* stream);
int svae_reduce_slabs_pair(const float* slabs0, float* out0, long long n0, long long stride0, const float* slabs1,
                           float* out1, long long n1, long long stride1, int nsplit, void* stream);
int svae_rowgemm_ld_prec(const float* x, int ldx, const void* frags, const float* bias, float* y, int R, int which,
                         int prec, void* stream);
int svae_mlp_prep_prec(const float* w1, const float* wl, const float* w2, void* frags, int prec, void* stream);
int svae_mlp_fwd_prec(const float* hs, const float* eps, const void* frags, const float* b1, const float* bl,
                      const float* b2, float* enc, float* tmp, float* z, float* d1, float* kl_row, int R, int prec,
                      void* stream);
int svae_mlp_bwd_prec(const float* dd1, const float* tmp, const float* eps, const float* enc, const float* w_kl,
                      const void* frags, float anneal, float* dtmp, float* denc, float* dhs, int R, int prec,
                      void* stream);
int svae_proj_prep_rows_range_split(const float* h2, int R, int R_pad, int row0, int row1, void* h2b, void* h2s,
                                    void* h2tb, long long part_stride, void* stream);
int svae_proj_prep_items_split(const float* w3, const float* b3, int I, int I_pad, void* w3b, void* w3s, void* w3tb,
                               long long part_stride, void* stream);
int svae_proj_nll_fwd_split(const void* h2s, const void* w3b, const float* h2, const float* w3, const float* b3,
                            const int* tgt, int R, int I, int K, float* part, float* nll_row, float* lse,
                            const int* sched, int nwg, int ns_q, int ns_rem, long long row_part_stride,
                            long long item_part_stride, void* stream);
int svae_proj_nll_fwd_dh_split(const void* h2s, void* h2b, const void* w3b, const void* w3tb, int I_pad,
                               const float* h2, const float* w3, const float* b3, const int* tgt, const float* w_nll,
                               int R, int I, int K, float* slabs, float* mpart, float* nll_row, float* lse, float* dd1,
                               const int* sched, int nwg, int ns_q, int ns_rem, int parts, long long row_part_stride,
                               long long item_part_stride, void* stream);
int svae_proj_nll_bwd_dw_split(const void* w3s, const void* h2b, const void* h2tb, int R, int R_pad, int I, int I_rows,
                               const float* h2, const float* w_nll, const int* occ_off, const int* occ_row,
                               const int* occ_items, int nocc_long, float* slabs, float* gw3, float* gb3,
                               const int* sched, int nwg, int ns_q, int ns_rem, int parts, long long row_part_stride,
                               long long item_part_stride, void* stream);

/* Timing hooks (measurement only; no counterpart in the reference): svae_proj_time_next arms the calling thread's next
 * sweep launch with start / stop events (from svae_event_create) attached to the dispatch itself; read the kernel's
 * duration with svae_event_elapsed_ms (synchronises on the stop event). */
/* ---- collectives for hosts without torch.distributed (thin RCCL wrappers taking the communicator) ----
 * No counterpart in the reference (single process).  librccl is loaded on first use.  svae_rccl_unique_id: 128 bytes
 * made by rank 0 and handed to every rank; svae_rccl_comm_create: one communicator per process and GPU (collective
 * over the nranks callers).  The exchanges are in place on the flat gradient / parameter buffers and are enqueued
 * on `stream`: the sum all-reduce of a bucket; for the sharded decoder.linear2 bucket (nranks shards of shard_count
 * floats) the reduce-scatter that leaves rank r's shard of the sum at bucket + r * shard_count and, after the
 * optimizer step over that shard, the all-gather of the parameter shards.  Errors: < 0 argument / loader errors,
 * 1000 + ncclResult_t for RCCL's own. */
int svae_rccl_unique_id(void* out128);
int svae_rccl_comm_create(void** comm, int nranks, int rank, const void* id128);
int svae_rccl_comm_destroy(void* comm);
int svae_allreduce_sum_f32(void* comm, float* buf, long long count, void* stream);
int svae_reduce_scatter_sum_f32(void* comm, float* bucket, long long shard_count, int rank, void* stream);
int svae_allgather_f32(void* comm, float* bucket, long long shard_count, int rank, void* stream);

int svae_event_create(void** out_event);
/* Ordering events between the step's streams (host helpers, no counterpart in the reference): an event without
 * timing, its record on a stream, a stream's wait for it. */
int svae_sync_event_create(void** out_event);
int svae_event_record(void* event, void* stream);
int svae_stream_wait_event(void* stream, void* event);
int svae_event_destroy(void* event);
int svae_event_elapsed_ms(void* start, void* stop, float* ms);
int svae_proj_time_next(void* start, void* stop);

/* ---- Adam with L2 weight decay on one flat buffer ----------------------------------------------------
 * Replaces optimizer.step() nb:1206 with torch.optim.Adam(weight_decay=5e-3) nb:1260-1263 (and the
 * zero_grad pair nb:1197-1198: gradients are written, not accumulated).  step is 1-based. */
int svae_adam_step(float* p, const float* g, float* m, float* v, long long n, int step, float lr, float b1, float b2,
                   float eps, float wd, float gscale, void* stream);

/* ---- the notebook's other optimizers on one flat buffer ---------------------------------------------
 * Replaces optimizer.step() nb:1206 for hyper_params['optimizer'] in {adagrad, adadelta, rmsprop}
 * (nb:1252-1259, nb:1264-1267), torch.optim semantics with L2 decay added to the gradient first:
 * kind 0 Adagrad  (a = state_sum, decay = lr_decay, eps 1e-10 by default), kind 1 Adadelta (a = square_avg,
 * b = acc_delta, decay = rho), kind 2 RMSprop (a = square_avg, decay = alpha; plain: no momentum, not
 * centered).  b may be NULL unless kind == 1; step is 1-based (Adagrad's lr decay).
 * decay is a double: 1 - decay is rounded from the double difference, as torch does. */
int svae_optimizer_step(int kind, float* p, const float* g, float* a, float* b, long long n, int step, float lr,
                        double decay, float eps, float wd, float gscale, void* stream);

/* ---- scoring ------------------------------------------------------------------------------------------
 * svae_masked_topk replaces the seen-item mask nb:621-624 and torch.sort(-scores)[:k] nb:632-636:
 * score_i = s_i - |s_i * m_i| * 1e8, descending, ties by lower index.  seen / seen_off: CSR of each
 * user's input items.  work: scratch [n_users][ld]. */
int svae_masked_topk(const float* scores, int ld, const int* seen, const int* seen_off, int n_users, int I, int K,
                     int* out_idx, float* out_val, float* work, void* stream);

/* svae_heldout_ranks replaces the hit search of evaluate() nb:636-645 (`argsorted[:k].cpu()`, then `movie in rec_list`
 * / `rec_list.index(movie)` per held-out item).  topk_idx [n_users][K]: svae_masked_topk's ids, users in the plan's
 * (length-sorted) order; row_key / last_row: the plan's (the user's global index is row_key[last_row[u]] >> 20);
 * held / held_off: CSR of the held-out items of ALL n_total_users users of the evaluation, by global index.
 * rank_out[j] (same indexing as held): 0-based position of held[j] in its user's list, -1 when absent.  Only the
 * entries of this call's users are written. */
int svae_heldout_ranks(const int* topk_idx, int n_users, int K, const long long* row_key, const int* last_row,
                       const int* held, const long long* held_off, long long n_total_users, int* rank_out,
                       void* stream);

/* No counterpart in the reference (single process).  The rebuild behind the sparse exchange of the item-embedding
 * gradient in data parallel runs: all_ids [world][cap] (int64, -1 = padding) and all_vals [world][cap][width] are the
 * all-gathered (row id, row) pairs; the rank's own rows (own_ids[0..n_own)) of table [rows][width] are zeroed, then the
 * ranks' rows are added in rank order (one launch per rank: the same additions in the same order on every rank).
 * ids are distinct within a rank's list; width % 4 == 0. */
int svae_sparse_rows_rebuild(float* table, long long rows, int width, const long long* all_ids, const float* all_vals,
                             int world, int cap, const long long* own_ids, int n_own, void* stream);

/* No counterpart in the reference.  svae_masked_topk for callers that CANNOT ALLOCATE the `work` buffer: code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  Same kernel and results; the scratch is
 * a static device array inside the library (32 MiB per GPU, held until the process ends): n_users * ld <= 8 Mi
 * floats (e.g. 256 users x 20 032, 8 users x 1 000 000), larger calls are refused (rc < 0).  No hipMalloc / hipFree /
 * allocating routine is called.  The stream must belong to the current device; svae_check_device() on that device
 * beforehand does the one-time address look-up (otherwise the first call does).  Calls that overlap on one GPU
 * share the array and race: keep them on one stream. */
int svae_masked_topk_static(const float* scores, int ld, const int* seen, const int* seen_off, int n_users, int I,
                            int K, int* out_idx, float* out_val, void* stream);

/* ---- batch plans (HOST functions, no GPU work) -------------------------------------------------------------
 * Replaces the per-user packing of DataReader.iter (nb:485-526: x [1,T] and the dense one-hot next-k targets
 * y_s [1,T,I] of nb:509-513) for a BATCH of users: time-major rows, users sorted by length, int32 target table with
 * duplicates inside a row removed (scatter_ semantics), per-row loss weights (nb:797, nb:803, nb:808), rows grouped by
 * input item, target occurrences grouped by item.  The dataset is CSR: items[user_off[u] .. user_off[u+1]) are user
 * u's item ids in time order (at least 2).  idx[B] = the batch's users, keys[B] = their ids for the counter-based
 * noise (NULL: position in the batch).  next_k as hyper_params['next_k'] (nb:88).  total_items = 0 leaves the
 * occurrence tables out.  Thread-safe (no shared state); meant to be called from worker threads.
 *   svae_plan_create   builds the plan in host memory and fills `info`; NULL on error (svae_last_error)
 *   svae_plan_pack     copies it as ONE packed buffer (every array at a multiple of 256 bytes, info.off[i] = byte
 *                      offset of array i in the order x, tgt [R][K], w_nll, w_kl, row_off [Tmax+1], nact [Tmax],
 *                      last_row [B], perm [R], seg_off [nseg+1], seg_item [nseg], row_key [R] (int64), seen [R],
 *                      seen_off [B+1], occ_off [I+1], occ_row [nocc], occ_items [I]) into dst -- e.g. pinned memory,
 *                      uploaded with one copy -- and the host-side tables order [B], lengths [B] (int64), row_off
 *   svae_plan_destroy  frees it */
typedef struct svae_plan_info {
    int B, R, Tmax, K, nseg, nseg_long, nocc, nocc_long, split_t, split_r, total_items, reserved;
    long long total_bytes;
    long long off[16];
} svae_plan_info;
void* svae_plan_create(const int* items, const long long* user_off, const long long* idx, const long long* keys,
                       int B, int next_k, int total_items, svae_plan_info* info);
int svae_plan_pack(void* plan, void* dst, long long dst_bytes, long long* order, long long* lengths, int* row_off);
void svae_plan_destroy(void* plan);

/* ---- helpers -------------------------------------------------------------------------------------------- */
int svae_colsum(const float* a, float* out, float* partials, int R, int N, int ld, int nparts, void* stream);
int svae_transpose(const float* in, float* out, int rows, int cols, void* stream);
int svae_gather_rows(const float* in, const int* idx, float* out, int nrows, int ld_in, int ld_out, int n,
                     void* stream);

/* Zero fill of nbytes at p on the stream (model.zero_grad() nb:1197 for the embedding gradient, the only gradient
 * that is accumulated into rather than written). */
int svae_zero(void* p, long long nbytes, void* stream);
/* One workgroup busy for usec microseconds: lets the host test whether two streams run concurrently (streams
 * that share a hardware queue do not).  No counterpart in the reference (single stream there). */
int svae_spin(int usec, void* stream);
/* nwg one-wave workgroups, each busy for usec microseconds and holding lds_bytes (<= 65536) of LDS: a launch whose
 * DISPATCH lasts many rounds while most of the chip stays free.  Lets the host test whether a second stream's
 * launches are held up behind it (queues that share a dispatcher pipe; engine.py, _pick_concurrent_stream).  No
 * counterpart in the reference. */
int svae_spin_grid(int usec, int nwg, int lds_bytes, void* stream);
/* buf[slot] = the device wall clock (100 MHz ticks) when the stream reaches this point (diagnostic: tools/step_stamps.py).
 * No counterpart in the reference. */
int svae_stamp(unsigned long long* buf, int slot, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SVAE_HIP_H */
