// Versa and Mahalanobis heads: the other two single-step classifiers of the reference README.
//   model/classifier_heads.py:121-180   VersaClassifier   (CNAPs): class means -> two DenseResidualBlock hyper-networks
//                                        (model/mlps.py:33-50) -> weight [C][D], bias [C] of a linear layer
//   model/classifier_heads.py:265-368   MahalanobisClassifier (Simple CNAPs): per-class mean + regularised covariance,
//                                        torch.inverse -> precisions; logits = -(mu_c - q)^T P_c (mu_c - q)
// Kernels: a few-rows dense layer (weights streamed once, one wave per output unit), masked covariance estimation,
// covariance blending, batched in-place blocked Gauss-Jordan inversion of the SPD matrices (cov + I), and the
// Mahalanobis quadratic form (difference -> MFMA GEMM through the 1x1 path of conv_igemm -> row dot).
#include "common.h"

namespace orbit {

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---- y[r][o] = act(x[r] . W[o] + b[o]) (+ residual[r][o]); R <= 16 rows, one wave per output unit -------------------
constexpr int DENSE_MAX_ROWS = 16;
__global__ __launch_bounds__(256) void dense_rows_kernel(const float* __restrict__ x, int R, int in,
                                                         const float* __restrict__ W, const float* __restrict__ b,
                                                         int out, int act, const float* __restrict__ residual,
                                                         float* __restrict__ y) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + wave;
    if (o >= out) return;
    float acc[DENSE_MAX_ROWS];
#pragma unroll
    for (int r = 0; r < DENSE_MAX_ROWS; ++r) acc[r] = 0.f;
    const float* w = W + (size_t)o * in;
    for (int k = lane; k < in; k += 64) {
        const float wk = w[k];
#pragma unroll
        for (int r = 0; r < DENSE_MAX_ROWS; ++r)
            if (r < R) acc[r] = fmaf(wk, x[(size_t)r * in + k], acc[r]);
    }
#pragma unroll
    for (int r = 0; r < DENSE_MAX_ROWS; ++r) {
        if (r >= R) break;
        float v = wave_sum64(acc[r]) + (b ? b[o] : 0.f);
        if (act == ORBIT_ACT_ELU) v = v > 0.f ? v : expm1f(v);
        else if (act == ORBIT_ACT_RELU) v = fmaxf(v, 0.f);
        if (residual) v += residual[(size_t)r * out + o];
        if (lane == 0) y[(size_t)r * out + o] = v;
    }
}

// ---- covariance ------------------------------------------------------------------------------------------------------
// cov[z][i][j] = sum over the rows of set z of (x_i - m_i)(x_j - m_j) / (n_z - 1); set z < C: rows with label
// class_ids[z]; set z == C: all rows. means [C+1][D], counts [C+1]. grid (D/32, D/32, C+1), 256 threads, 32x32 tile.
__global__ __launch_bounds__(256) void cov_kernel(const float* __restrict__ feats, const int64_t* __restrict__ labels,
                                                  const int64_t* __restrict__ class_ids, int N, int D, int C,
                                                  const float* __restrict__ means, const float* __restrict__ counts,
                                                  float* __restrict__ cov) {
    __shared__ float Xi[32][33], Xj[32][33];
    const int z = blockIdx.z;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // ty 0..7
    const float* mu = means + (size_t)z * D;
    const int64_t cid = z < C ? class_ids[z] : 0;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int n0 = 0; n0 < N; n0 += 32) {
        // stage 32 rows x 32 columns of both column ranges, centred and masked
        for (int r = ty; r < 32; r += 8) {
            const int n = n0 + r;
            const bool sel = n < N && (z == C || labels[n] == cid);
            const int ci = i0 + tx, cj = j0 + tx;
            Xi[r][tx] = (sel && ci < D) ? feats[(size_t)n * D + ci] - mu[ci] : 0.f;
            Xj[r][tx] = (sel && cj < D) ? feats[(size_t)n * D + cj] - mu[cj] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int r = 0; r < 32; ++r) {
            const float xj = Xj[r][tx];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = fmaf(Xi[r][ty + 8 * q], xj, acc[q]);
        }
        __syncthreads();
    }
    const float cnt = counts[z];
    const float f = cnt > 1.5f ? 1.0f / (cnt - 1.0f) : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = i0 + ty + 8 * q, j = j0 + tx;
        if (i < D && j < D) cov[((size_t)z * D + i) * D + j] = acc[q] * f;
    }
}

// the reference's single-example branch (classifier_heads.py:361-364): examples [1][D] are centred by THEIR OWN mean
// over the features and the "covariance" is the scalar sum(e^2)/(D-1), later broadcast over the whole matrix
__global__ __launch_bounds__(256) void cov_single_kernel(const float* __restrict__ feats,
                                                         const int64_t* __restrict__ labels,
                                                         const int64_t* __restrict__ class_ids, int N, int D, int C,
                                                         const float* __restrict__ counts, float* __restrict__ scalar) {
    __shared__ float red[256];
    const int z = blockIdx.x;
    scalar[z] = 0.f;
    if (counts[z] > 1.5f) return;
    int row = -1;
    for (int n = 0; n < N; ++n)
        if (labels[n] == class_ids[z]) {
            row = n;
            break;
        }
    if (row < 0) return;
    const float* x = feats + (size_t)row * D;
    float s = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) s += x[d];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    const float m = red[0] / (float)D;
    __syncthreads();
    float q = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) q += (x[d] - m) * (x[d] - m);
    red[threadIdx.x] = q;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) scalar[z] = red[0] / (float)(D - 1);
}

// sigma[z] = lambda_z cov_z + (1 - lambda_z) cov_task + I  (z < C), sigma[C] = cov_task + I   (:288, :306-310)
__global__ __launch_bounds__(256) void cov_blend_kernel(const float* __restrict__ cov, const float* __restrict__ counts,
                                                        const float* __restrict__ scalar, int D, int C,
                                                        float* __restrict__ sigma) {
    const size_t DD = (size_t)D * D;
    const size_t total = DD * (C + 1);
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int z = (int)(e / DD);
        const size_t ij = e - (size_t)z * DD;
        const int i = (int)(ij / D), j = (int)(ij - (size_t)i * D);
        const float task = cov[(size_t)C * DD + ij];
        float v;
        if (z == C) {
            v = task;
        } else {
            const float n = counts[z];
            const float lam = n / (n + 1.0f);
            const float cz = n > 1.5f ? cov[e] : scalar[z];
            v = lam * cz + (1.0f - lam) * task;
        }
        sigma[e] = v + (i == j ? 1.0f : 0.f);
    }
}

// ---- batched in-place blocked Gauss-Jordan inverse (no pivoting: the inputs are covariance + identity, SPD) -----------
constexpr int GJ_NB = 32;

// step 1: dinv = inverse of the diagonal block k (unblocked Gauss-Jordan in LDS), one block per matrix
__global__ __launch_bounds__(256) void gj_diag_kernel(const float* __restrict__ A, int n, int k, float* __restrict__ dinv) {
    __shared__ float M[GJ_NB][GJ_NB + 1];
    const float* a = A + (size_t)blockIdx.x * n * n;
    const int k0 = k * GJ_NB, nb = min(GJ_NB, n - k0);
    for (int e = threadIdx.x; e < GJ_NB * GJ_NB; e += 256) {
        const int i = e / GJ_NB, j = e % GJ_NB;
        M[i][j] = (i < nb && j < nb) ? a[(size_t)(k0 + i) * n + k0 + j] : (i == j ? 1.f : 0.f);
    }
    __syncthreads();
    for (int p = 0; p < nb; ++p) {
        const float piv = 1.0f / M[p][p];
        __syncthreads();
        // row p scaled, column p eliminated from the other rows (in-place Gauss-Jordan)
        for (int e = threadIdx.x; e < GJ_NB * GJ_NB; e += 256) {
            const int i = e / GJ_NB, j = e % GJ_NB;
            if (i >= nb || j >= nb || i == p || j == p) continue;
            M[i][j] -= M[i][p] * piv * M[p][j];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < GJ_NB; e += 256) {
            if (e >= nb || e == p) continue;
            M[p][e] *= piv;       // row p
        }
        __syncthreads();
        for (int e = threadIdx.x; e < GJ_NB; e += 256) {
            if (e >= nb || e == p) continue;
            M[e][p] *= -piv;      // column p
        }
        if (threadIdx.x == 0) M[p][p] = piv;
        __syncthreads();
    }
    float* d = dinv + (size_t)blockIdx.x * GJ_NB * GJ_NB;
    for (int e = threadIdx.x; e < GJ_NB * GJ_NB; e += 256) d[e] = M[e / GJ_NB][e % GJ_NB];
}

// step 2: row panel A[K][J] <- dinv . A[K][J] for every column block J != k; grid (n/NB, batch)
__global__ __launch_bounds__(256) void gj_row_kernel(float* __restrict__ A, int n, int k, const float* __restrict__ dinv) {
    __shared__ float Dm[GJ_NB][GJ_NB + 1], Bm[GJ_NB][GJ_NB + 1];
    const int jb = blockIdx.x;
    if (jb == k) return;
    float* a = A + (size_t)blockIdx.y * n * n;
    const float* d = dinv + (size_t)blockIdx.y * GJ_NB * GJ_NB;
    const int k0 = k * GJ_NB, j0 = jb * GJ_NB, nbk = min(GJ_NB, n - k0), nbj = min(GJ_NB, n - j0);
    for (int e = threadIdx.x; e < GJ_NB * GJ_NB; e += 256) {
        const int i = e / GJ_NB, j = e % GJ_NB;
        Dm[i][j] = d[e];
        Bm[i][j] = (i < nbk && j < nbj) ? a[(size_t)(k0 + i) * n + j0 + j] : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < GJ_NB * GJ_NB; e += 256) {
        const int i = e / GJ_NB, j = e % GJ_NB;
        if (i >= nbk || j >= nbj) continue;
        float s = 0.f;
#pragma unroll 8
        for (int t = 0; t < GJ_NB; ++t) s = fmaf(Dm[i][t], Bm[t][j], s);
        a[(size_t)(k0 + i) * n + j0 + j] = s;
    }
}

// step 3: A[I][J] -= A[I][K] . A[K][J] for I != k, J != k; grid (n/NB, n/NB, batch)
__global__ __launch_bounds__(256) void gj_update_kernel(float* __restrict__ A, int n, int k) {
    __shared__ float L[GJ_NB][GJ_NB + 1], U[GJ_NB][GJ_NB + 1];
    const int jb = blockIdx.x, ib = blockIdx.y;
    if (ib == k || jb == k) return;
    float* a = A + (size_t)blockIdx.z * n * n;
    const int k0 = k * GJ_NB, i0 = ib * GJ_NB, j0 = jb * GJ_NB;
    const int nbk = min(GJ_NB, n - k0), nbi = min(GJ_NB, n - i0), nbj = min(GJ_NB, n - j0);
    for (int e = threadIdx.x; e < GJ_NB * GJ_NB; e += 256) {
        const int r = e / GJ_NB, c = e % GJ_NB;
        L[r][c] = (r < nbi && c < nbk) ? a[(size_t)(i0 + r) * n + k0 + c] : 0.f;
        U[r][c] = (r < nbk && c < nbj) ? a[(size_t)(k0 + r) * n + j0 + c] : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < GJ_NB * GJ_NB; e += 256) {
        const int r = e / GJ_NB, c = e % GJ_NB;
        if (r >= nbi || c >= nbj) continue;
        float s = 0.f;
#pragma unroll 8
        for (int t = 0; t < GJ_NB; ++t) s = fmaf(L[r][t], U[t][c], s);
        a[(size_t)(i0 + r) * n + j0 + c] -= s;
    }
}

// step 4: column panel A[I][K] <- -A[I][K] . dinv for I != k, A[K][K] <- dinv; grid (n/NB, batch)
__global__ __launch_bounds__(256) void gj_col_kernel(float* __restrict__ A, int n, int k, const float* __restrict__ dinv) {
    __shared__ float Dm[GJ_NB][GJ_NB + 1], Bm[GJ_NB][GJ_NB + 1];
    const int ib = blockIdx.x;
    float* a = A + (size_t)blockIdx.y * n * n;
    const float* d = dinv + (size_t)blockIdx.y * GJ_NB * GJ_NB;
    const int k0 = k * GJ_NB, i0 = ib * GJ_NB, nbk = min(GJ_NB, n - k0), nbi = min(GJ_NB, n - i0);
    for (int e = threadIdx.x; e < GJ_NB * GJ_NB; e += 256) {
        const int i = e / GJ_NB, j = e % GJ_NB;
        Dm[i][j] = d[e];
        Bm[i][j] = (i < nbi && j < nbk) ? a[(size_t)(i0 + i) * n + k0 + j] : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < GJ_NB * GJ_NB; e += 256) {
        const int i = e / GJ_NB, j = e % GJ_NB;
        if (i >= nbi || j >= nbk) continue;
        float v;
        if (ib == k) {
            v = Dm[i][j];
        } else {
            float s = 0.f;
#pragma unroll 8
            for (int t = 0; t < GJ_NB; ++t) s = fmaf(Bm[i][t], Dm[t][j], s);
            v = -s;
        }
        a[(size_t)(i0 + i) * n + k0 + j] = v;
    }
}

int launch_spd_inverse_inplace(float* A, int n, int batch, float* dinv_scratch, hipStream_t s) {
    const int nblk = cdiv(n, GJ_NB);
    for (int k = 0; k < nblk; ++k) {
        gj_diag_kernel<<<batch, 256, 0, s>>>(A, n, k, dinv_scratch);
        gj_row_kernel<<<dim3(nblk, batch), 256, 0, s>>>(A, n, k, dinv_scratch);
        gj_update_kernel<<<dim3(nblk, nblk, batch), 256, 0, s>>>(A, n, k);
        gj_col_kernel<<<dim3(nblk, batch), 256, 0, s>>>(A, n, k, dinv_scratch);
    }
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

// ---- Mahalanobis predict helpers ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maha_diff_kernel(const float* __restrict__ Q, const float* __restrict__ mu, int M,
                                                        int D, float* __restrict__ diff) {
    const size_t total = (size_t)M * D;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256)
        diff[e] = mu[e % D] - Q[e];
}

// logits[m][c] = -scale * sum_j first[m][j] * diff[m][j]; one wave per row
__global__ __launch_bounds__(256) void maha_rowdot_kernel(const float* __restrict__ first, const float* __restrict__ diff,
                                                          int M, int D, int C, int c, float scale,
                                                          float* __restrict__ logits) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + wave;
    if (m >= M) return;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s = fmaf(first[(size_t)m * D + d], diff[(size_t)m * D + d], s);
    s = wave_sum64(s);
    if (lane == 0) logits[(size_t)m * C + c] = -scale * s;
}

// psym = P + P^T (the gradient of d^T P d is (P + P^T) d; the stored precisions are only numerically symmetric)
__global__ __launch_bounds__(256) void symmetrise_kernel(const float* __restrict__ P, int D, float* __restrict__ out) {
    const size_t total = (size_t)D * D;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t i = e / D, j = e - i * D;
        out[e] = P[e] + P[j * D + i];
    }
}

// dq[m][:] (+)= scale * dlogits[m][c] * g[m][:]
__global__ __launch_bounds__(256) void maha_bwd_acc_kernel(const float* __restrict__ g, const float* __restrict__ dlogits,
                                                           int M, int D, int C, int c, float scale,
                                                           float* __restrict__ dq) {
    const size_t total = (size_t)M * D;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t m = e / D;
        const float v = scale * dlogits[m * C + c] * g[e];
        dq[e] = c == 0 ? v : dq[e] + v;
    }
}

__global__ __launch_bounds__(256) void task_mean_kernel(const float* __restrict__ sums, const float* __restrict__ counts,
                                                        int C, int D, float* __restrict__ means,
                                                        float* __restrict__ cnt_out) {
    // means[z<C] = sums[z] / counts[z]; means[C] = sum_z sums[z] / sum_z counts[z]; cnt_out[C+1]
    const int d = blockIdx.x * 256 + threadIdx.x;
    float total = 0.f, n = 0.f;
    for (int z = 0; z < C; ++z) {
        const float cz = counts[z];
        if (d < D) {
            const float sz = sums[(size_t)z * D + d];
            means[(size_t)z * D + d] = sz / fmaxf(cz, 1.f);
            total += sz;
        }
        n += cz;
        if (d == 0) cnt_out[z] = cz;
    }
    if (d < D) means[(size_t)C * D + d] = total / fmaxf(n, 1.f);
    if (d == 0) cnt_out[C] = n;
}

}  // namespace orbit

using namespace orbit;

extern "C" {

int orbit_dense_rows(const float* x, int R, int in, const float* W, const float* b, int out, int act,
                     const float* residual, float* y, orbit_stream_t stream) {
    ORBIT_REQUIRE(x && W && y, "dense_rows: null pointer");
    ORBIT_REQUIRE(R > 0 && R <= DENSE_MAX_ROWS && in > 0 && out > 0, "dense_rows: bad sizes (rows must be 1..%d)",
                  DENSE_MAX_ROWS);
    dense_rows_kernel<<<cdiv(out, 4), 256, 0, (hipStream_t)stream>>>(x, R, in, W, b, out, act, residual, y);
    ORBIT_LAUNCH_CHECK();
    return ORBIT_OK;
}

int orbit_spd_inverse(const float* A, float* Ainv, int n, int batch, orbit_stream_t stream) {
    ORBIT_REQUIRE(A && Ainv && n > 0 && batch > 0, "spd_inverse: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (A != Ainv) ORBIT_HIP_CHECK(hipMemcpyAsync(Ainv, A, (size_t)batch * n * n * sizeof(float), hipMemcpyDeviceToDevice, s));
    float* scratch = nullptr;
    ORBIT_HIP_CHECK(hipMallocAsync(reinterpret_cast<void**>(&scratch), (size_t)batch * GJ_NB * GJ_NB * sizeof(float), s));
    const int rc = launch_spd_inverse_inplace(Ainv, n, batch, scratch, s);
    (void)hipFreeAsync(scratch, s);
    return rc;
}

size_t orbit_mahalanobis_workspace_bytes(int N, int M, int D, int C) {
    (void)N;
    const size_t DD = (size_t)D * D;
    size_t fl = (size_t)(C + 1) * D      /* means incl. task mean */
                + 2 * (size_t)(C + 1)     /* counts, single-example scalars */
                + (size_t)C * D + C       /* per-class sums / counts of the configure kernel */
                + 2 * (size_t)(C + 1) * DD    /* covariance estimates + the blended matrices that are inverted in place */
                + (size_t)(C + 1) * GJ_NB * GJ_NB;
    const size_t predict = 2 * (size_t)M * D + DD + conv_packed_floats(D, D, 1, 1, 0);
    if (predict > fl) fl = predict;
    return align_up(fl * sizeof(float), 256) + 256;
}

int orbit_mahalanobis_configure(const float* features, const int64_t* labels, const int64_t* class_ids, int N, int D,
                                int C, float* means, float* task_mean, float* precisions, float* task_precision,
                                void* workspace, size_t workspace_bytes, orbit_stream_t stream) {
    ORBIT_REQUIRE(features && labels && class_ids && means && task_mean && precisions && task_precision && workspace,
                  "mahalanobis_configure: null pointer");
    ORBIT_REQUIRE(N > 1 && D > 1 && C > 0, "mahalanobis_configure: bad sizes");
    ORBIT_REQUIRE(workspace_bytes >= orbit_mahalanobis_workspace_bytes(N, 1, D, C), "mahalanobis_configure: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const size_t DD = (size_t)D * D;
    float* ws = static_cast<float*>(workspace);
    float* mu = ws;                                 // [C+1][D]
    float* cnt = mu + (size_t)(C + 1) * D;          // [C+1]
    float* scalar = cnt + (C + 1);                  // [C+1]
    float* sums = scalar + (C + 1);                 // [C][D]
    float* counts = sums + (size_t)C * D;           // [C]
    float* cov = counts + C;                        // [C+1][D][D]
    cov = reinterpret_cast<float*>(align_up(reinterpret_cast<uintptr_t>(cov), 16));
    float* sigma = cov + (size_t)(C + 1) * DD;       // [C+1][D][D] blended covariances, inverted in place
    float* dinv = sigma + (size_t)(C + 1) * DD;
    // per-class sums / counts in ascending row order (the prototype kernel), then means
    if (int rc = orbit_proto_configure(features, labels, class_ids, 1, N, 1, D, C, sums, counts, stream)) return rc;
    task_mean_kernel<<<cdiv(D, 256), 256, 0, s>>>(sums, counts, C, D, mu, cnt);
    ORBIT_LAUNCH_CHECK();
    const int nb = cdiv(D, 32);
    cov_kernel<<<dim3(nb, nb, C + 1), 256, 0, s>>>(features, labels, class_ids, N, D, C, mu, cnt, cov);
    ORBIT_LAUNCH_CHECK();
    cov_single_kernel<<<C, 256, 0, s>>>(features, labels, class_ids, N, D, C, cnt, scalar);
    ORBIT_LAUNCH_CHECK();
    // sigma lives in the caller's workspace: a stream-ordered allocation here (hipMallocAsync / hipFreeAsync of 6-26 MB per
    // task, rounds 1-2) falls back to real allocations and frees once the pool has been trimmed, and a free synchronises
    // the device - the host then sat out the whole task (host enqueue 16.9 of 17.35 ms in the simple-CNAPs bench line)
    cov_blend_kernel<<<4096, 256, 0, s>>>(cov, cnt, scalar, D, C, sigma);
    hipError_t e = hipGetLastError();
    int rc = ORBIT_OK;
    if (e != hipSuccess) rc = set_err(ORBIT_ERR_HIP, "mahalanobis_configure: %s", hipGetErrorString(e));
    if (rc == ORBIT_OK) rc = launch_spd_inverse_inplace(sigma, D, C + 1, dinv, s);
    if (rc == ORBIT_OK) {
        e = hipMemcpyAsync(precisions, sigma, (size_t)C * DD * sizeof(float), hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess)
            e = hipMemcpyAsync(task_precision, sigma + (size_t)C * DD, DD * sizeof(float), hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(means, mu, (size_t)C * D * sizeof(float), hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess)
            e = hipMemcpyAsync(task_mean, mu + (size_t)C * D, (size_t)D * sizeof(float), hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) rc = set_err(ORBIT_ERR_HIP, "mahalanobis_configure: %s", hipGetErrorString(e));
    }
    return rc;
}

int orbit_mahalanobis_predict(const float* features, const float* means, const float* precisions, int M, int D, int C,
                              float logit_scale, float* logits, void* workspace, size_t workspace_bytes,
                              orbit_stream_t stream) {
    ORBIT_REQUIRE(features && means && precisions && logits && workspace, "mahalanobis_predict: null pointer");
    ORBIT_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && C > 0, "mahalanobis_predict: bad sizes (D must be a multiple of 4)");
    ORBIT_REQUIRE(workspace_bytes >= orbit_mahalanobis_workspace_bytes(2, M, D, C), "mahalanobis_predict: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const size_t DD = (size_t)D * D;
    float* ws = static_cast<float*>(workspace);
    float* diff = ws;                        // [M][D]
    float* first = diff + (size_t)M * D;     // [M][D]
    float* pt = first + (size_t)M * D;       // [D][D] precision transposed (so that first = diff . P exactly)
    float* packed = pt + DD;
    for (int c = 0; c < C; ++c) {
        int blocks = (int)(((size_t)M * D + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        maha_diff_kernel<<<blocks, 256, 0, s>>>(features, means + (size_t)c * D, M, D, diff);
        ORBIT_LAUNCH_CHECK();
        if (int rc = launch_transpose(precisions + (size_t)c * DD, pt, D, D, s)) return rc;
        if (int rc = conv_pack_weights(pt, packed, D, D, 1, 1, 0, s)) return rc;
        ConvDesc d;
        d.x = diff, d.w_packed = packed, d.y = first, d.scale = d.shift = d.residual = d.gate = nullptr;
        d.B = M, d.H = 1, d.W = 1, d.Cin = D, d.Cout = D, d.KH = 1, d.KW = 1, d.stride = 1, d.pad_t = 0, d.pad_l = 0;
        d.Ho = 1, d.Wo = 1, d.act = ORBIT_ACT_NONE, d.pool2 = 0, d.x_nchw = 0;
        if (int rc = launch_conv(d, s)) return rc;
        maha_rowdot_kernel<<<cdiv(M, 4), 256, 0, s>>>(first, diff, M, D, C, c, logit_scale, logits);
        ORBIT_LAUNCH_CHECK();
    }
    return ORBIT_OK;
}

/* d(features) of orbit_mahalanobis_predict: dq_m = scale * sum_c dlogits[m][c] (P_c + P_c^T)(mu_c - q_m) */
int orbit_mahalanobis_predict_backward(const float* dlogits, const float* features, const float* means,
                                       const float* precisions, int M, int D, int C, float logit_scale, float* dfeatures,
                                       void* workspace, size_t workspace_bytes, orbit_stream_t stream) {
    ORBIT_REQUIRE(dlogits && features && means && precisions && dfeatures && workspace,
                  "mahalanobis_predict_backward: null pointer");
    ORBIT_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && C > 0, "mahalanobis_predict_backward: bad sizes");
    ORBIT_REQUIRE(workspace_bytes >= orbit_mahalanobis_workspace_bytes(2, M, D, C),
                  "mahalanobis_predict_backward: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const size_t DD = (size_t)D * D;
    float* ws = static_cast<float*>(workspace);
    float* diff = ws;
    float* g = diff + (size_t)M * D;
    float* psym = g + (size_t)M * D;
    float* packed = psym + DD;
    int blocks = (int)(((size_t)M * D + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    for (int c = 0; c < C; ++c) {
        maha_diff_kernel<<<blocks, 256, 0, s>>>(features, means + (size_t)c * D, M, D, diff);
        symmetrise_kernel<<<4096, 256, 0, s>>>(precisions + (size_t)c * DD, D, psym);
        ORBIT_LAUNCH_CHECK();
        if (int rc = conv_pack_weights(psym, packed, D, D, 1, 1, 0, s)) return rc;
        ConvDesc d;
        d.x = diff, d.w_packed = packed, d.y = g, d.scale = d.shift = d.residual = d.gate = nullptr;
        d.B = M, d.H = 1, d.W = 1, d.Cin = D, d.Cout = D, d.KH = 1, d.KW = 1, d.stride = 1, d.pad_t = 0, d.pad_l = 0;
        d.Ho = 1, d.Wo = 1, d.act = ORBIT_ACT_NONE, d.pool2 = 0, d.x_nchw = 0;
        if (int rc = launch_conv(d, s)) return rc;
        maha_bwd_acc_kernel<<<blocks, 256, 0, s>>>(g, dlogits, M, D, C, c, logit_scale, dfeatures);
        ORBIT_LAUNCH_CHECK();
    }
    return ORBIT_OK;
}

}  // extern "C"
