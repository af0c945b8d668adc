// Discrete-time path (BASELINE config 5): dense symmetric-normalised adjacency for GCNConv and the
// TGCN gate arithmetic (tgm/nn/encoder/tgcn.py:8-157).  Snapshot graphs on this path are small
// (tgbn-trade: 255 nodes), so A_hat is materialised densely and the propagation A_hat (X W) runs on
// the exact-fp32 MFMA GEMM (sgemm_nt) like every other dense contraction here.
// GCNConv itself is third-party to the reference (torch_geometric); implemented from its published
// definition (gcn_norm with add_remaining_self_loops): parity unpinned upstream.
#include "common.h"

namespace tgmx {

__global__ __launch_bounds__(256) void gcn_init_kernel(float* deg, float* loop_w, float* A, long long N, long long ld) {
  const long long step = (long long)gridDim.x * blockDim.x;
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long long i = i0; i < N; i += step) {
    deg[i] = 0.f;
    reinterpret_cast<int*>(loop_w)[i] = -1;  // index of the node's last explicit self-loop edge
  }
  for (long long i = i0; i < N * ld; i += step) A[i] = 0.f;
}

// existing self loops replace the fill value (add_remaining_self_loops); other edges add to the in-degree
template <typename IdxT>
__global__ __launch_bounds__(256) void gcn_degree_kernel(const IdxT* src, const IdxT* dst, const float* w, long long E,
                                                         float* deg, float* loop_w, int add_self_loops) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  // an explicit self loop replaces the fill value; with several on one node the LAST edge wins
  // (the sequential semantics of add_remaining_self_loops' index assignment)
  if (add_self_loops && src[e] == dst[e]) atomicMax(reinterpret_cast<int*>(loop_w) + dst[e], (int)e);
  else atomicAdd(&deg[dst[e]], w ? w[e] : 1.f);
}

__global__ __launch_bounds__(256) void gcn_dinv_kernel(float* deg, const float* loop_w, const float* w, float fill, float* A,
                                                       long long N, long long ld, int add_self_loops) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int le = reinterpret_cast<const int*>(loop_w)[i];
  const float lw = !add_self_loops ? 0.f : (le < 0 ? fill : (w ? w[le] : 1.f));
  const float d = deg[i] + lw;
  const float dinv = d > 0.f ? 1.0f / sqrtf(d) : 0.f;
  deg[i] = dinv;
  if (add_self_loops) A[i * ld + i] = dinv * lw * dinv;
}

// A[dst, src] += dinv[src] * w * dinv[dst]   (messages flow src -> dst, aggregated at dst)
template <typename IdxT>
__global__ __launch_bounds__(256) void gcn_fill_kernel(const IdxT* src, const IdxT* dst, const float* w, long long E,
                                                       const float* dinv, float* A, long long ld, int add_self_loops) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const long long s = src[e], d = dst[e];
  if (s == d && add_self_loops) return;  // folded into the diagonal above
  atomicAdd(&A[d * ld + s], dinv[s] * (w ? w[e] : 1.f) * dinv[d]);
}

// out[i, :C] = a[i, :C];  out[i, C:2C] = b[i, :C] * (gate ? sigmoid(gate[i, :C]) : 1)
__global__ __launch_bounds__(256) void tgcn_concat_kernel(const float* a, long long lda, const float* b, const float* gate, int C,
                                                          long long N, float* out) {
  const long long total = N * 2 * C;
  const long long step = (long long)gridDim.x * blockDim.x;
  for (long long x = (long long)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += step) {
    const long long i = x / (2 * C);
    const int c = (int)(x - i * 2 * C);
    float v;
    if (c < C) v = a[i * lda + c];
    else {
      v = b[i * C + (c - C)];
      if (gate) v *= 1.f / (1.f + expf(-gate[i * C + (c - C)]));
    }
    out[x] = v;
  }
}

// H' = U * H + (1 - U) * tanh(c_pre),  U = sigmoid(u_pre)      (tgcn.py:151-156)
__global__ __launch_bounds__(256) void tgcn_output_kernel(const float* u_pre, const float* c_pre, const float* H, long long n, float* out) {
  const long long step = (long long)gridDim.x * blockDim.x;
  for (long long x = (long long)blockIdx.x * blockDim.x + threadIdx.x; x < n; x += step) {
    const float u = 1.f / (1.f + expf(-u_pre[x]));
    out[x] = u * H[x] + (1.f - u) * tanhf(c_pre[x]);
  }
}

// ---- backward of the cell (training: examples/nodeproppred/tgcn.py:92 calls loss.backward() through tgcn.py:151-156) ----------------
// out = U H + (1 - U) Cc,  U = sigmoid(u_pre), Cc = tanh(c_pre):
//   du_pre = dout (H - Cc) U (1 - U),   dc_pre = dout (1 - U) (1 - Cc^2),   dH = dout U   (the direct term; the gates' terms are added below)
__global__ __launch_bounds__(256) void tgcn_gate_backward_kernel(const float* __restrict__ dout, const float* __restrict__ u_pre,
                                                                 const float* __restrict__ c_pre, const float* __restrict__ H, long long n,
                                                                 float* __restrict__ du_pre, float* __restrict__ dc_pre, float* __restrict__ dH) {
  const long long step = (long long)gridDim.x * blockDim.x;
  for (long long x = (long long)blockIdx.x * blockDim.x + threadIdx.x; x < n; x += step) {
    const float u = 1.f / (1.f + expf(-u_pre[x]));
    const float cc = tanhf(c_pre[x]);
    const float g = dout[x];
    du_pre[x] = g * (H[x] - cc) * u * (1.f - u);
    dc_pre[x] = g * (1.f - u) * (1.f - cc * cc);
    dH[x] = g * u;
  }
}

// The candidate gate read [conv_c(X) | H R] (R = sigmoid(r_pre)): with dcat_c = dc_pre W_c its right half is d(H R), so
//   dr_pre = d(HR) H R (1 - R),   dH += d(HR) R  (+ the right halves of dcat_u = du_pre W_u and, when given, dcat_r = dr_pre W_r: the
// update and reset gates read H itself).  dcat_* are [N, 2C]; a NULL one is skipped; dr_pre NULL: only the accumulation.
__global__ __launch_bounds__(256) void tgcn_reset_backward_kernel(const float* __restrict__ dcat_c, const float* __restrict__ dcat_u,
                                                                  const float* __restrict__ dcat_r, const float* __restrict__ r_pre,
                                                                  const float* __restrict__ H, int C, long long N, float* __restrict__ dr_pre,
                                                                  float* __restrict__ dH) {
  const long long total = N * C;
  const long long step = (long long)gridDim.x * blockDim.x;
  for (long long x = (long long)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += step) {
    const long long i = x / C;
    const int c = (int)(x - i * C);
    const long long right = i * 2 * C + C + c;
    float acc = dH[x];
    if (dcat_c) {
      const float r = 1.f / (1.f + expf(-r_pre[x]));
      const float dhr = dcat_c[right];
      if (dr_pre) dr_pre[x] = dhr * H[x] * r * (1.f - r);
      acc += dhr * r;
    }
    if (dcat_u) acc += dcat_u[right];
    if (dcat_r) acc += dcat_r[right];
    dH[x] = acc;
  }
}

}  // namespace tgmx

using namespace tgmx;

extern "C" int tgmx_tgcn_gate_backward(const float* dout, const float* u_pre, const float* c_pre, const float* H, int64_t n, float* du_pre,
                                       float* dc_pre, float* dH, tgmx_stream_t stream) {
  TGMX_REQUIRE(n >= 0, "tgcn_gate_backward: bad size");
  if (n == 0) return TGMX_OK;
  TGMX_REQUIRE(dout && u_pre && c_pre && H && du_pre && dc_pre && dH, "tgcn_gate_backward: null pointer");
  long long blocks = (n + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(tgcn_gate_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dout, u_pre, c_pre, H, (long long)n, du_pre,
                     dc_pre, dH);
  TGMX_CHECK_LAUNCH("tgcn_gate_backward");
  return TGMX_OK;
}

extern "C" int tgmx_tgcn_reset_backward(const float* dcat_c, const float* dcat_u, const float* dcat_r, const float* r_pre, const float* H, int32_t C,
                                        int64_t N, float* dr_pre, float* dH, tgmx_stream_t stream) {
  TGMX_REQUIRE(C > 0 && N >= 0, "tgcn_reset_backward: bad sizes");
  if (N == 0) return TGMX_OK;
  TGMX_REQUIRE(dH && (!dcat_c || (r_pre && H)), "tgcn_reset_backward: null pointer");
  long long blocks = (N * C + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(tgcn_reset_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dcat_c, dcat_u, dcat_r, r_pre, H, C,
                     (long long)N, dr_pre, dH);
  TGMX_CHECK_LAUNCH("tgcn_reset_backward");
  return TGMX_OK;
}

template <typename IdxT>
static int gcn_norm_dense_impl(const IdxT* src, const IdxT* dst, const float* weight, int64_t E, int64_t N, float fill, int32_t add_self_loops,
                               float* A, int64_t ld, float* workspace, tgmx_stream_t stream) {
  TGMX_REQUIRE(E >= 0 && N > 0 && ld >= N, "gcn_norm_dense: bad sizes");
  TGMX_REQUIRE(A && workspace && (E == 0 || (src && dst)), "gcn_norm_dense: null pointer");
  hipStream_t st = (hipStream_t)stream;
  float* deg = workspace;
  float* loop_w = workspace + N;
  long long blocks = (N * ld + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(gcn_init_kernel, dim3((unsigned)blocks), dim3(256), 0, st, deg, loop_w, A, (long long)N, (long long)ld);
  if (E) hipLaunchKernelGGL(gcn_degree_kernel<IdxT>, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, src, dst, weight, (long long)E, deg, loop_w, add_self_loops);
  hipLaunchKernelGGL(gcn_dinv_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, deg, loop_w, weight, fill, A, (long long)N, (long long)ld, add_self_loops);
  if (E) hipLaunchKernelGGL(gcn_fill_kernel<IdxT>, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, src, dst, weight, (long long)E, deg, A, (long long)ld, add_self_loops);
  TGMX_CHECK_LAUNCH("gcn_norm_dense");
  return TGMX_OK;
}

extern "C" int tgmx_gcn_norm_dense(const int64_t* src, const int64_t* dst, const float* weight, int64_t E, int64_t N, float fill,
                                   int32_t add_self_loops, float* A, int64_t ld, float* workspace, tgmx_stream_t stream) {
  return gcn_norm_dense_impl<int64_t>(src, dst, weight, E, N, fill, add_self_loops, A, ld, workspace, stream);
}

extern "C" int tgmx_tgcn_concat(const float* a, int64_t lda, const float* b, const float* gate_pre, int32_t C, int64_t N, float* out,
                                tgmx_stream_t stream) {
  TGMX_REQUIRE(C > 0 && N >= 0 && lda >= C, "tgcn_concat: bad sizes");
  if (N == 0) return TGMX_OK;
  TGMX_REQUIRE(a && b && out, "tgcn_concat: null pointer");
  long long blocks = (N * 2 * C + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(tgcn_concat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, (long long)lda, b, gate_pre, C, (long long)N, out);
  TGMX_CHECK_LAUNCH("tgcn_concat");
  return TGMX_OK;
}

extern "C" int tgmx_tgcn_output(const float* u_pre, const float* c_pre, const float* H, int64_t n, float* out, tgmx_stream_t stream) {
  TGMX_REQUIRE(n >= 0, "tgcn_output: bad size");
  if (n == 0) return TGMX_OK;
  TGMX_REQUIRE(u_pre && c_pre && H && out, "tgcn_output: null pointer");
  long long blocks = (n + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(tgcn_output_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, u_pre, c_pre, H, (long long)n, out);
  TGMX_CHECK_LAUNCH("tgcn_output");
  return TGMX_OK;
}

// the cell's inference forward as one call (tgm_amd/nn/tgcn.py TGCN.forward composes the same entry points one ctypes call at a time)
extern "C" int tgmx_tgcn_forward(const tgmx_tgcn_fwd_t* a, tgmx_stream_t stream) {
  TGMX_REQUIRE(a, "tgcn_forward: null argument block");
  const int64_t N = a->N;
  const int C = a->C;
  TGMX_REQUIRE(N > 0 && C > 0 && a->in_ch > 0 && a->ldA >= N, "tgcn_forward: bad sizes");
  TGMX_REQUIRE(a->x && a->W3 && a->b3 && a->H && a->A && a->norm_ws && a->xwt && a->G && a->cat && a->out, "tgcn_forward: null pointer");
  for (int g = 0; g < 3; ++g) TGMX_REQUIRE(a->lin_w[g] && a->lin_b[g] && a->pre[g], "tgcn_forward: null pointer (gate %d)", g);
  // (idx32: the loader's batches carry int32 endpoints -- read as they are instead of through a widened copy)
  int rc = a->idx32 ? gcn_norm_dense_impl<int32_t>(reinterpret_cast<const int32_t*>(a->src), reinterpret_cast<const int32_t*>(a->dst), a->edge_w, a->E, N,
                                                   a->fill, a->add_self_loops, a->A, a->ldA, a->norm_ws, stream)
                    : tgmx_gcn_norm_dense(a->src, a->dst, a->edge_w, a->E, N, a->fill, a->add_self_loops, a->A, a->ldA, a->norm_ws, stream);
  if (rc) return rc;
  // (X W3^T)^T = W3 X^T: [3C, N];  G = A_hat (X W3^T) + b3: [N, 3C]
  if ((rc = tgmx_sgemm_nt(a->W3, a->in_ch, a->x, a->in_ch, a->xwt, N, 3 * C, (int32_t)N, a->in_ch, nullptr, 0, 1, 0, 0, 0, stream))) return rc;
  if ((rc = tgmx_sgemm_nt(a->A, a->ldA, a->xwt, N, a->G, 3 * C, N, 3 * C, (int32_t)N, a->b3, 0, 1, 0, 0, 0, stream))) return rc;
  for (int g = 0; g < 3; ++g) {  // u, r, c: the candidate's input is gated by the reset gate's pre-activation
    if ((rc = tgmx_tgcn_concat(a->G + g * C, 3 * C, a->H, g == 2 ? a->pre[1] : nullptr, C, N, a->cat, stream))) return rc;
    if ((rc = tgmx_sgemm_nt(a->cat, 2 * C, a->lin_w[g], 2 * C, a->pre[g], C, N, C, 2 * C, a->lin_b[g], 0, 1, 0, 0, 0, stream))) return rc;
  }
  return tgmx_tgcn_output(a->pre[0], a->pre[2], a->H, N * C, a->out, stream);
}
