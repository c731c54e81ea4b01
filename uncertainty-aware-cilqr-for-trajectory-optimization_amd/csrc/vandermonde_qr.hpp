// vandermonde_qr.hpp — the polynomial fit of the reference's LocalPlanner::polyfit (I/LocalPlanner.cpp:101-117),
// written once for the host pre-step (local_planner.cpp) and the batched device pre-step (local_plan.hip).
//
// The reference solves the Vandermonde system with Eigen's `colPivHouseholderQr().solve(y)` (:114).  The same
// algorithm is written out here: Householder QR with column pivoting on the largest remaining column norm, rank
// cut-off at max-column-norm²·eps²/rows scaled by the remaining rows, Qᵀy, back-substitution on the leading
// nonzero-pivot block, zeros for the cut columns.  Every sum runs in the order Eigen's kernels run it for these sizes,
// and floating-point contraction is off, so host and device produce the same bits from the same Vandermonde matrix.
//
// `Store` supplies the working storage (host: plain arrays; device: one LDS slot per lane):
//   double& m(i, j)      rows×cols matrix, in: V(i,j) = x_i^j, out: the QR factors
//   double& c(i)         rows vector, in: y, out: scratch
//   double& col_norm(j)  cols
//   double& tau(k)       min(rows, cols)
//   int&    swap_with(k) min(rows, cols)
//   int&    order(j)     cols
#pragma once

#include <float.h>
#include <math.h>

#if defined(__HIPCC__)
#define CILQR_HD __host__ __device__ inline
#else
#define CILQR_HD inline
#endif

namespace cilqr {

// Row loops.  RM = 0: plain loops (host).  RM > 0 (device; the store has room for RM rows): fully unrolled over RM rows with
// the body told whether its row is in [from, rows) — loads are then unconditional and independent of one another, so a lane
// has RM of them in flight instead of waiting out one LDS round trip per row.  Sums run in the same order either way.
template <int RM, class Body>
CILQR_HD void for_rows(int from, int rows, Body&& body) {
  if constexpr (RM > 0) {
#pragma unroll
    for (int i = 0; i < RM; ++i) body(i, i >= from && i < rows);
  } else {
    for (int i = from; i < rows; ++i) body(i, true);
  }
}

template <int RM, class Store>
CILQR_HD double tail_sq_norm(Store& s, int rows, int col, int from) {
#pragma clang fp contract(off)
  double acc = 0.0;
  for_rows<RM>(from, rows, [&](int i, bool on) {
#pragma clang fp contract(off)
    const double v = s.m(i, col);
    if (on) acc += v * v;
  });
  return acc;
}

template <class T>
CILQR_HD void swap_values(T& a, T& b) {
  const T t = a;
  a = b;
  b = t;
}

// Least squares min |V c - y|; coeffs[0..cols) receives the solution.
template <int RM = 0, class Store>
CILQR_HD void vandermonde_lstsq(Store& s, int rows, int cols, double* coeffs) {
#pragma clang fp contract(off)
  const int diag = rows < cols ? rows : cols;
  for (int k = 0; k < diag; ++k) {
    s.tau(k) = 0.0;
    s.swap_with(k) = 0;
  }
  double max_norm = 0.0;
  for (int j = 0; j < cols; ++j) {
    s.col_norm(j) = tail_sq_norm<RM>(s, rows, j, 0);
    if (j == 0 || s.col_norm(j) > max_norm) max_norm = s.col_norm(j);
  }
  const double cut = max_norm * (DBL_EPSILON * DBL_EPSILON) / (double)rows;
  int rank = diag;
  for (int k = 0; k < diag; ++k) {
    int pivot = k;
    for (int j = k + 1; j < cols; ++j)
      if (s.col_norm(j) > s.col_norm(pivot)) pivot = j;
    const double exact = tail_sq_norm<RM>(s, rows, pivot, k);
    s.col_norm(pivot) = exact;
    if (rank == diag && exact < cut * (double)(rows - k)) rank = k;
    s.swap_with(k) = pivot;
    if (pivot != k) {
      for_rows<RM>(0, rows, [&](int i, bool on) {
        const double a = s.m(i, k), b = s.m(i, pivot);
        if (on) { s.m(i, k) = b; s.m(i, pivot) = a; }
      });
      swap_values(s.col_norm(k), s.col_norm(pivot));
    }
    // Householder vector for column k (stored below the diagonal, unit leading entry implied)
    const double below = tail_sq_norm<RM>(s, rows, k, k + 1);
    const double head = s.m(k, k);
    double beta;
    if (below == 0.0) {
      s.tau(k) = 0.0;
      beta = head;
      for_rows<RM>(k + 1, rows, [&](int i, bool on) { if (on) s.m(i, k) = 0.0; });
    } else {
      beta = sqrt(head * head + below);
      if (head >= 0.0) beta = -beta;
      const double den = head - beta;
      for_rows<RM>(k + 1, rows, [&](int i, bool on) {
        const double v = s.m(i, k);
        if (on) s.m(i, k) = v / den;
      });
      s.tau(k) = (beta - head) / beta;
    }
    s.m(k, k) = beta;
    const double tk = s.tau(k);
    // reflect the trailing columns
    if constexpr (RM > 0) {
      // device: the Householder vector is read once per k and each trailing column once per (k, j), RM independent loads at
      // a time, into registers; the arithmetic and its order are those of the plain loops below
      double vk[RM];
#pragma unroll
      for (int i = 0; i < RM; ++i) vk[i] = s.m(i, k);
      for (int j = k + 1; j < cols; ++j) {
        if (rows - k == 1) {
          s.m(k, j) *= (1 - tk);
        } else {
          double bj[RM];
#pragma unroll
          for (int i = 0; i < RM; ++i) bj[i] = s.m(i, j);
          double dot = 0.0;
#pragma unroll
          for (int i = 0; i < RM; ++i)
            if (i >= k + 1 && i < rows) dot += vk[i] * bj[i];
          dot += s.m(k, j);
          s.m(k, j) -= tk * dot;
#pragma unroll
          for (int i = 0; i < RM; ++i)
            if (i >= k + 1 && i < rows) s.m(i, j) = bj[i] - tk * vk[i] * dot;
        }
        s.col_norm(j) -= s.m(k, j) * s.m(k, j);
      }
    } else {
      for (int j = k + 1; j < cols; ++j) {
        if (rows - k == 1) {
          s.m(k, j) *= (1 - tk);
        } else {
          double dot = 0.0;
          for (int i = k + 1; i < rows; ++i) dot += s.m(i, k) * s.m(i, j);
          dot += s.m(k, j);
          s.m(k, j) -= tk * dot;
          for (int i = k + 1; i < rows; ++i) s.m(i, j) -= tk * s.m(i, k) * dot;
        }
        s.col_norm(j) -= s.m(k, j) * s.m(k, j);
      }
    }
  }
  for (int j = 0; j < cols; ++j) s.order(j) = j;
  for (int k = 0; k < diag; ++k) swap_values(s.order(k), s.order(s.swap_with(k)));

  for (int j = 0; j < cols; ++j) coeffs[j] = 0.0;
  if (rank == 0) return;
  for (int k = 0; k < rank; ++k) {  // c ← H_k c
    const double tk = s.tau(k);
    if (rows - k == 1) {
      s.c(k) *= (1 - tk);
      continue;
    }
    if constexpr (RM > 0) {
      double vk[RM], cv[RM];
#pragma unroll
      for (int i = 0; i < RM; ++i) {
        vk[i] = s.m(i, k);
        cv[i] = s.c(i);
      }
      double dot = 0.0;
#pragma unroll
      for (int i = 0; i < RM; ++i)
        if (i >= k + 1 && i < rows) dot += vk[i] * cv[i];
      dot += s.c(k);
      s.c(k) -= tk * dot;
#pragma unroll
      for (int i = 0; i < RM; ++i)
        if (i >= k + 1 && i < rows) s.c(i) = cv[i] - tk * vk[i] * dot;
    } else {
      double dot = 0.0;
      for (int i = k + 1; i < rows; ++i) dot += s.m(i, k) * s.c(i);
      dot += s.c(k);
      s.c(k) -= tk * dot;
      for (int i = k + 1; i < rows; ++i) s.c(i) -= tk * s.m(i, k) * dot;
    }
  }
  for (int i = rank - 1; i >= 0; --i) {
    double acc = s.c(i);
    for (int j = i + 1; j < rank; ++j) acc -= s.m(i, j) * s.c(j);
    s.c(i) = acc / s.m(i, i);
  }
  for (int i = 0; i < rank; ++i) coeffs[s.order(i)] = s.c(i);
}

}  // namespace cilqr
