// TGN training: backward of the hand-written forward kernels in tgn.hip (the dense projections differentiate through
// sgemm_nt / sgemm_tn / colsum from tgat.hip / tgat_bwd.hip).  The reference trains through torch autograd
// (examples/linkproppred/tgn.py:97-118); memory and last_update are buffers (no gradient), so the parameters reached
// are the shared Time2Vec, the GRU cell and the TransformerConv projections.
#include "common.h"

namespace tgmx {

// d(gi), d(gh) of GRUCell's gate arithmetic:  out = (1 - z) n + z h,  r = s(gi_r + gh_r), z = s(gi_z + gh_z),
// n = tanh(gi_n + r gh_n)                                                        (h is a buffer row: no dh)
__global__ __launch_bounds__(256) void tgn_gru_gate_backward_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                                                    const float* __restrict__ h, const float* __restrict__ dout,
                                                                    int M, long long R, float* __restrict__ dgi,
                                                                    float* __restrict__ dgh) {
  const long long total = R * M;
  const long long step = (long long)gridDim.x * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
    const long long r = e / M;
    const int c = (int)(e - r * M);
    const float* gir = gi + r * 3 * M;
    const float* ghr = gh + r * 3 * M;
    const float rg = 1.f / (1.f + expf(-(gir[c] + ghr[c])));
    const float zg = 1.f / (1.f + expf(-(gir[M + c] + ghr[M + c])));
    const float ghn = ghr[2 * M + c];
    const float ng = tanhf(gir[2 * M + c] + rg * ghn);
    const float g = dout[e];
    const float dn = g * (1.f - zg);
    const float dz = g * (h[e] - ng);
    const float dan = dn * (1.f - ng * ng);
    const float dar = dan * ghn * rg * (1.f - rg);
    const float daz = dz * zg * (1.f - zg);
    float* dgir = dgi + r * 3 * M;
    float* dghr = dgh + r * 3 * M;
    dgir[c] = dar;
    dghr[c] = dar;
    dgir[M + c] = daz;
    dghr[M + c] = daz;
    dgir[2 * M + c] = dan;
    dghr[2 * M + c] = dan * rg;
  }
}

// Aggregation backward: only the Time2Vec columns of a message depend on parameters.  Per row the same event
// selection as tgn_aggregate_kernel (LAST: largest float32(t), first in walk order on ties; MEAN: all events / count);
// part[row, t] = d(tw[t]) contribution, part[row, T + t] = d(tb[t]) contribution (summed over rows by tgmx_colsum).
struct AggrBwdArgs {
  // per ROW snapshots taken at forward time (the reference's training loop calls update_state, which rewrites the
  // per-node windows and last_update, BEFORE loss.backward(); the event log itself is append-only within an epoch)
  const int64_t* row_lo[2];   // [R] first log position of the row's source- / destination-role window
  const int32_t* row_cnt[2];  // [R]
  const int64_t* row_lu;      // [R] last_update of the row's node at forward time
  const int64_t* log_t;
  const float* tw;
  const float* tb;
  const float* d_aggr;  // [R, W]
  float* part;          // [R, 2T]
  long long R;
  int W, toff, T, mean;
};

__global__ __launch_bounds__(256) void tgn_aggregate_backward_kernel(const AggrBwdArgs a) {
  const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= a.R) return;
  const int lane = lane_id();
  float* out = a.part + row * 2LL * a.T;
  const long long lu_v = a.row_lu[row];
  const long long lo0 = a.row_lo[0][row], lo1 = a.row_lo[1][row];
  const int c0 = a.row_cnt[0][row], c1 = a.row_cnt[1][row];
  const int total = c0 + c1;
  if (total == 0) {
    for (int c = lane; c < 2 * a.T; c += kWave) out[c] = 0.f;
    return;
  }
  auto pos = [&](int i) -> long long { return i < c0 ? lo0 + i : lo1 + (i - c0); };
  float fbest = -__builtin_inff();
  int ibest = 0x7fffffff;
  if (!a.mean) {
    for (int i = lane; i < total; i += kWave) {
      const float f = (float)a.log_t[pos(i)];
      if (f > fbest) {
        fbest = f;
        ibest = i;
      }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const float f2 = __shfl_xor(fbest, o);
      const int i2 = __shfl_xor(ibest, o);
      if (f2 > fbest || (f2 == fbest && i2 < ibest)) {
        fbest = f2;
        ibest = i2;
      }
    }
  }
  const float* g = a.d_aggr + row * (long long)a.W + a.toff;
  for (int t = lane; t < a.T; t += kWave) {
    float dw = 0.f, db = 0.f;
    const float gt = g[t], w = a.tw[t], b = a.tb[t];
    if (!a.mean) {
      const float dt = (float)(a.log_t[pos(ibest)] - lu_v);
      const float gs = -sin_t2v(__fmaf_rn(dt, w, b)) * gt;
      dw = gs * dt;
      db = gs;
    } else {
      for (int i = 0; i < total; ++i) {
        const float dt = (float)(a.log_t[pos(i)] - lu_v);
        const float gs = -sin_t2v(__fmaf_rn(dt, w, b)) * gt;
        dw = __fmaf_rn(gs, dt, dw);
        db += gs;
      }
      dw /= (float)total;
      db /= (float)total;
    }
    out[t] = dw;
    out[a.T + t] = db;
  }
}

// edge_attr[e, :T] = cos(fma(float(lu[src[e]] - t[e]), w, b)):  part[e, t] = d(tw[t]), part[e, T + t] = d(tb[t])
__global__ __launch_bounds__(256) void tconv_edge_attr_backward_kernel(const int64_t* __restrict__ lu_local, const int64_t* __restrict__ src,
                                                                       const int64_t* __restrict__ t, const float* __restrict__ tw,
                                                                       const float* __restrict__ tb, const float* __restrict__ d_attr,
                                                                       int T, int W, long long E, float* __restrict__ part) {
  const long long total = E * T;
  const long long step = (long long)gridDim.x * blockDim.x;
  for (long long x = (long long)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += step) {
    const long long e = x / T;
    const int c = (int)(x - e * T);
    const float dt = (float)(lu_local[src[e]] - t[e]);
    const float gs = -sin_t2v(__fmaf_rn(dt, tw[c], tb[c])) * d_attr[e * W + c];
    part[e * 2LL * T + c] = gs * dt;
    part[e * 2LL * T + T + c] = gs;
  }
}

// TransformerConv attention backward (forward: tconv_attend_kernel).  One wave per target node i, head by head, lanes
// over the head's C channels (C <= 64):
//   pass 1: running max m, normaliser l and the attention output o = sum_p alpha_p val_p  (as the forward);
//   delta = dout . o;   pass 2 per incoming edge p: alpha_p, dalpha_p = dout . val_p, ds_p = alpha_p (dalpha_p - delta),
//     dq_i += ds_p scale (k_j + e_p);   dk_j += ds_p scale q_i;   dv_j += alpha_p dout;   de_p = ds_p scale q_i + alpha_p dout
// dk / dv rows are shared between targets: float atomics (summation order is not fixed: ~1 ulp run-to-run).
struct TconvBwdArgs {
  const float* q;
  const float* k;
  const float* v;
  const float* eproj;
  const int64_t* order;
  const int64_t* src;
  const int64_t* seg_lo;
  const int64_t* seg_hi;
  const float* dout;  // [U, H*C]
  float* dq;          // [U, H*C] written
  float* dk;          // [U, H*C] zero-initialised, accumulated
  float* dv;          // [U, H*C] zero-initialised, accumulated
  float* de;          // [E, H*C] written for every edge (each edge has exactly one target)
  long long U;
  int H, C;
  float scale;
  DropoutArgs drop;  // the forward's dropout on the attention coefficients, regenerated: element e * H + h
};

__global__ __launch_bounds__(256) void tconv_attend_backward_kernel(const TconvBwdArgs a) {
  const long long i = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= a.U) return;
  const int lane = lane_id();
  const int HC = a.H * a.C;
  const long long lo = a.seg_lo[i], hi = a.seg_hi[i];
  const bool on = lane < a.C;
  for (int h = 0; h < a.H; ++h) {
    const int col = h * a.C + lane;
    const float qi = on ? a.q[i * HC + col] : 0.f;
    const float go = on ? a.dout[i * HC + col] : 0.f;
    if (hi <= lo) {
      if (on) a.dq[i * HC + col] = 0.f;
      continue;
    }
    float m = -__builtin_inff(), l = 0.f, acc = 0.f;
    for (long long p = lo; p < hi; ++p) {
      const long long e = a.order[p], j = a.src[e];
      const float ke = on ? a.k[j * HC + col] + a.eproj[e * HC + col] : 0.f;
      float part = qi * ke;
      for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
      const float s = part * a.scale;
      const float mn = s > m ? s : m;
      const float corr = expf(m - mn), w = expf(s - mn);
      const float val = on ? a.v[j * HC + col] + a.eproj[e * HC + col] : 0.f;
      acc = acc * corr + w * dropout_scale(a.drop, (unsigned long long)e * a.H + h) * val;
      l = l * corr + w;
      m = mn;
    }
    float delta = go * (acc / l);  // sum_e alpha_e d alpha_e with d alpha_e = mk_e (go . val_e): the dropped output, again
    for (int o = 32; o > 0; o >>= 1) delta += __shfl_xor(delta, o);
    float dqi = 0.f;
    for (long long p = lo; p < hi; ++p) {
      const long long e = a.order[p], j = a.src[e];
      const float ep = on ? a.eproj[e * HC + col] : 0.f;
      const float ke = on ? a.k[j * HC + col] + ep : 0.f;
      const float val = on ? a.v[j * HC + col] + ep : 0.f;
      float part = qi * ke, dal = go * val;
      for (int o = 32; o > 0; o >>= 1) {
        part += __shfl_xor(part, o);
        dal += __shfl_xor(dal, o);
      }
      const float alpha = expf(part * a.scale - m) / l;
      const float mk = dropout_scale(a.drop, (unsigned long long)e * a.H + h);
      const float ds = alpha * (dal * mk - delta) * a.scale;  // softmax backward on the pre-dropout coefficients
      dqi = __fmaf_rn(ds, ke, dqi);
      if (on) {
        const float ad = alpha * mk;  // the coefficient the value actually got
        atomicAdd(&a.dk[j * HC + col], ds * qi);
        atomicAdd(&a.dv[j * HC + col], ad * go);
        a.de[e * HC + col] = ds * qi + ad * go;
      }
    }
    if (on) a.dq[i * HC + col] = dqi;
  }
}

}  // namespace tgmx

using namespace tgmx;

extern "C" int tgmx_tgn_gru_gate_backward(const float* gi, const float* gh, const float* h, const float* dout, int32_t M, int64_t R,
                                          float* dgi, float* dgh, tgmx_stream_t stream) {
  TGMX_REQUIRE(M > 0 && R >= 0, "tgn_gru_gate_backward: bad sizes");
  if (R == 0) return TGMX_OK;
  TGMX_REQUIRE(gi && gh && h && dout && dgi && dgh, "tgn_gru_gate_backward: null pointer");
  long long blocks = (R * M + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(tgn_gru_gate_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gi, gh, h, dout, M,
                     (long long)R, dgi, dgh);
  TGMX_CHECK_LAUNCH("tgn_gru_gate_backward");
  return TGMX_OK;
}

extern "C" int tgmx_tgn_aggregate_backward(int64_t R, const int64_t* row_lo_s, const int32_t* row_cnt_s, const int64_t* row_lo_d,
                                           const int32_t* row_cnt_d, const int64_t* row_last_update, const int64_t* log_t, int32_t M,
                                           int32_t D, const float* tw, const float* tb, int32_t T, int32_t mean, const float* d_aggr,
                                           float* part, tgmx_stream_t stream) {
  TGMX_REQUIRE(R >= 0 && M > 0 && D >= 0 && T > 0, "tgn_aggregate_backward: bad sizes");
  if (R == 0) return TGMX_OK;
  TGMX_REQUIRE(row_lo_s && row_cnt_s && row_lo_d && row_cnt_d && row_last_update && tw && tb && d_aggr && part,
               "tgn_aggregate_backward: null pointer");
  AggrBwdArgs a{};
  a.row_lo[0] = row_lo_s; a.row_lo[1] = row_lo_d; a.row_cnt[0] = row_cnt_s; a.row_cnt[1] = row_cnt_d; a.row_lu = row_last_update;
  a.log_t = log_t; a.tw = tw; a.tb = tb; a.d_aggr = d_aggr; a.part = part; a.R = R; a.W = 2 * M + D + T; a.toff = 2 * M + D; a.T = T;
  a.mean = mean;
  hipLaunchKernelGGL(tgn_aggregate_backward_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  TGMX_CHECK_LAUNCH("tgn_aggregate_backward");
  return TGMX_OK;
}

extern "C" int tgmx_tconv_edge_attr_backward(const int64_t* last_update_local, const int64_t* src, const int64_t* t, const float* tw,
                                             const float* tb, const float* d_attr, int32_t T, int32_t D, int64_t E, float* part,
                                             tgmx_stream_t stream) {
  TGMX_REQUIRE(T > 0 && D >= 0 && E >= 0, "tconv_edge_attr_backward: bad sizes");
  if (E == 0) return TGMX_OK;
  TGMX_REQUIRE(last_update_local && src && t && tw && tb && d_attr && part, "tconv_edge_attr_backward: null pointer");
  long long blocks = (E * T + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(tconv_edge_attr_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, last_update_local, src,
                     t, tw, tb, d_attr, T, T + D, (long long)E, part);
  TGMX_CHECK_LAUNCH("tconv_edge_attr_backward");
  return TGMX_OK;
}

extern "C" int tgmx_tconv_attend_backward(const float* q, const float* k, const float* v, const float* eproj, const int64_t* order,
                                          const int64_t* src, const int64_t* seg_lo, const int64_t* seg_hi, int64_t U, int32_t H,
                                          int32_t C, float scale, const float* dout, float* dq, float* dk, float* dv, float* de,
                                          const tgmx_dropout_t* drop, tgmx_stream_t stream) {
  TGMX_REQUIRE(U >= 0 && H > 0 && C > 0 && C <= 64, "tconv_attend_backward: bad sizes (C <= 64)");
  if (U == 0) return TGMX_OK;
  TGMX_REQUIRE(q && k && v && eproj && order && src && seg_lo && seg_hi && dout && dq && dk && dv && de, "tconv_attend_backward: null pointer");
  TconvBwdArgs a{q, k, v, eproj, order, src, seg_lo, seg_hi, dout, dq, dk, dv, de, U, H, C, scale};
  a.drop = make_dropout(drop);
  hipLaunchKernelGGL(tconv_attend_backward_kernel, dim3((unsigned)((U + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  TGMX_CHECK_LAUNCH("tconv_attend_backward");
  return TGMX_OK;
}
