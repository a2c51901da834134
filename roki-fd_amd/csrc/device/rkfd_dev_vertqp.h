/* rkfd_dev_vertqp.h - part of the device code of the batched rkFDUpdate step (see rkfd_device.h):
 * the Vert plugin's rigid branch - friction pyramids, the objective q = A'A + L, c = A'c and the
 * active-set QP solver (reference src/rkfd_vert.c:73-103,235-283, src/rkfd_opt_qp.c:43-181).
 * Included by rkfd_device.h only; compiles for gfx950 and under the lane emulator. */
#ifndef RKFD_DEV_VERTQP_H
#define RKFD_DEV_VERTQP_H

/* ------------------------------------------------------------------------ */
/* wave-cooperative dense kernels on small SPD systems held in LDS (n <= 64, row-major, leading
 * dimension ld).  All lanes call them (uniform control flow); lane i owns row / entry i. */

/* These routines only touch the lower triangle; pk = true: it is stored packed by rows (entry (r,c) at r(r+1)/2 + c,
 * ld unused), which is how the factor of Q is kept - half the LDS of a full square. */
#define RKFD_WI(r,c) ( pk ? ( ( (r)*( (r)+1 ) ) >> 1 ) + (c) : (r)*ld + (c) )
/* in-place Cholesky: lower triangle <- factor, diagonal <- 1/L_ii (left-looking; the pivot of
 * column k travels by v_readlane) */
template<bool pk> RKFD_DEV void rkfd_w_chol(double *Mx, int ld, int n)
{
  const int lane = LANE();
  for( int k=0; k<n; k++ ){
    double s = 0;
    if( lane >= k && lane < n ){
      s = Mx[RKFD_WI( lane, k )];
      /* (unrolled: the loads of several terms go out together - the loop is bound by the latency of LDS, not by its arithmetic;
       *  the order of the subtractions stays) */
#pragma unroll 8
      for( int j=0; j<k; j++ ) s -= Mx[RKFD_WI( lane, j )]*Mx[RKFD_WI( k, j )];
    }
    const double rinv = RKFD_RCP( sqrt( BCAST( s, k ) ) );
    if( lane == k ) Mx[RKFD_WI( k, k )] = rinv;
    else if( lane > k && lane < n ) Mx[RKFD_WI( lane, k )] = s*rinv;
    LDS_FENCE();
  }
}
/* y = L^-1 b : lane i passes b_i and receives y_i */
template<bool pk> RKFD_DEV double rkfd_w_fwd(const double *Mx, int ld, int n, double bi)
{
  const int lane = LANE();
  double s = bi, yi = 0;
  for( int j=0; j<n; j++ ){
    const double yj = BCAST( s, j )*Mx[RKFD_WI( j, j )];
    if( lane == j ) yi = yj;
    if( lane > j && lane < n ) s -= Mx[RKFD_WI( lane, j )]*yj;
  }
  return yi;
}
/* x = L^-T y */
template<bool pk> RKFD_DEV double rkfd_w_back(const double *Mx, int ld, int n, double yi)
{
  const int lane = LANE();
  double s = yi, xi = 0;
  for( int j=n-1; j>=0; j-- ){
    const double xj = BCAST( s, j )*Mx[RKFD_WI( j, j )];
    if( lane == j ) xi = xj;
    if( lane < j ) s -= Mx[RKFD_WI( j, lane )]*xj;
  }
  return xi;
}
#undef RKFD_WI
/* entry (r, c <= r) of the packed factor of Q */
#define RKFD_QI(r,c) ( ( ( (r)*( (r)+1 ) ) >> 1 ) + (c) )

/* ---- the factor of Q in REGISTERS -----------------------------------------------------------------------------------------
 * One step under the Vert plugin is a dependent chain, not a throughput problem: at ONE instance per CU it takes as long as at
 * seven (profiles/r03_vert_qp.txt), and most of that chain were the triangular solves with the factor of Q read from LDS term by
 * term - the forward substitution for W = L^-1 C' with lane = column walked all n(n+1)/2 entries one after the other whatever the
 * number of columns (15 k cycles per KKT solve for typically 0 .. 2 columns).  With at most RKFD_QP_NQ = 24 unknowns (8 contact
 * vertices) lane i keeps row i of L (for L^-1) and column i (for L^-T) in registers for the whole QP; a substitution is then
 * n steps of { multiply, v_readlane, fma } with nothing to wait for, and a column of W costs one such pass - two columns share
 * a pass (independent chains in the same instructions).  The arithmetic per entry - operands and order - is that of the routines
 * above, so the results are the same bits.  Worlds with a larger capacity, and the kernels not compiled for one world, keep the
 * LDS routines. */
#ifndef RKFD_QP_NQ
#  if defined(RKFD_SPEC)
#    define RKFD_QP_NQ ( RKFD_SPEC_VERT_RIGID == 2 ? 3*RKFD_SPEC_MAXRG : 0 )      /* (the world's own bound) */
#  else
#    define RKFD_QP_NQ RKFD_QP_NQ_MAX
#  endif
#endif
template<int NQ> struct rkfdQpFactorT {
  double Lr[NQ > 0 ? NQ : 1];      /* L[lane][j], j < lane (else 0) */
  double Lt[NQ > 0 ? NQ : 1];      /* L[j][lane], j > lane (else 0) */
  double rd;                       /* 1 / L[lane][lane] */
};
typedef rkfdQpFactorT<RKFD_QP_NQ> rkfdQpFactor;
template<int NQ> RKFD_DEV void rkfd_qreg_load(rkfdQpFactorT<NQ> &F, const double *Q, int n)
{
  const int lane = LANE();
  const int base = ( lane*( lane+1 ) ) >> 1;
#pragma unroll
  for( int j=0; j<NQ; j++ ){
    F.Lr[j] = ( j < lane && lane < n ) ? Q[base + j] : 0.0;
    F.Lt[j] = ( j > lane && j < n ) ? Q[RKFD_QI( j, lane )] : 0.0;
  }
  F.rd = lane < n ? Q[base + lane] : 0.0;
}
/* The Cholesky factorisation itself with lane = row in registers, right-looking: after the pivot of column k every later entry
 * ( i, c ) of the lower triangle takes its term - l_ik l_ck, so an entry receives its terms in ascending k exactly as the
 * left-looking rkfd_w_chol subtracts them (same operands, same order: same bits); the values a step needs from another lane come
 * by v_readlane, nothing goes through LDS until the factor is stored at the end (packed, for the transposed read of
 * rkfd_qreg_load).  24 k cycles of LDS round trips per QP became 6 k. */
template<int NQ> RKFD_DEV void rkfd_qreg_chol(double *Q, int n)
{
  const int lane = LANE();
  const int base = ( lane*( lane+1 ) ) >> 1;
  double R[NQ > 0 ? NQ : 1];
#pragma unroll
  for( int j=0; j<NQ; j++ ) R[j] = ( j <= lane && lane < n ) ? Q[base + j] : 0.0;
#pragma unroll
  for( int k=0; k<NQ; k++ ){
    if( k < n ){
      const double rinv = RKFD_RCP( sqrt( BCAST( R[k], k ) ) );
      R[k] = lane == k ? rinv : R[k]*rinv;
#pragma unroll
      for( int c=k+1; c<NQ; c++ ) R[c] = fma( -R[k], BCAST( R[k], c ), R[c] );
    }
  }
#pragma unroll
  for( int j=0; j<NQ; j++ ) if( j <= lane && lane < n ) Q[base + j] = R[j];
}
/* y = L^-1 b for two right-hand sides at once (lane i passes b_i, receives y_i); rows above j0 are known to be zero in both */
template<int NQ> RKFD_DEV void rkfd_qreg_fwd2(const rkfdQpFactorT<NQ> &F, int n, int j0, double &a, double &b)
{
  const int lane = LANE();
  double sa = a, sb = b, ya = 0, yb = 0;
#pragma unroll
  for( int j=0; j<NQ; j++ ){
    if( j >= j0 && j < n ){
      const double pa = BCAST( sa*F.rd, j ), pb = BCAST( sb*F.rd, j );
      if( lane == j ){ ya = pa; yb = pb; }
      sa = fma( -F.Lr[j], pa, sa ); sb = fma( -F.Lr[j], pb, sb );
    }
  }
  a = ya; b = yb;
}
template<int NQ> RKFD_DEV double rkfd_qreg_fwd(const rkfdQpFactorT<NQ> &F, int n, double bi)
{
  const int lane = LANE();
  double s = bi, yi = 0;
#pragma unroll
  for( int j=0; j<NQ; j++ ){
    if( j < n ){
      const double yj = BCAST( s*F.rd, j );
      if( lane == j ) yi = yj;
      s = fma( -F.Lr[j], yj, s );
    }
  }
  return yi;
}
/* x = L^-T y */
template<int NQ> RKFD_DEV double rkfd_qreg_back(const rkfdQpFactorT<NQ> &F, int n, double yi)
{
  const int lane = LANE();
  double s = yi, xi = 0;
#pragma unroll
  for( int j=NQ-1; j>=0; j-- ){
    if( j < n ){
      const double xj = BCAST( s*F.rd, j );
      if( lane == j ) xi = xj;
      s = fma( -F.Lt[j], xj, s );
    }
  }
  return xi;
}
/* ( L' v )_lane: v_lane / rd + sum over j > lane of L[j][lane] v_j, the terms in ascending j */
template<int NQ> RKFD_DEV double rkfd_qreg_ltv(const rkfdQpFactorT<NQ> &F, int n, double vi)
{
  double u = vi/F.rd;
#pragma unroll
  for( int j=0; j<NQ; j++ ) if( j < n ) u = fma( F.Lt[j], BCAST( vi, j ), u );
  return u;
}

#ifndef RKFD_EMU
/* A Gram product G'G (r x r, r <= 32) on the matrix cores: the Vert QP's Q = A'A (once per evaluation, on by default) and the
 * Schur complement S = W'W of its iterations (switch RKFD_VERT_MFMA_S, measured: no gain); see DESIGN.md "MFMA".  W is n x r in LDS (row stride ldq); v_mfma_f64_16x16x4_f64 takes A[i = lane & 15][k = lane >> 4] and
 * B[k = lane >> 4][j = lane & 15] - for a Gram product both operands are W[k][column], so a lane loads ONE value per column block
 * and k-step where the VALU loop loads 2 n values per entry.  Tiles (0,0), (0,1), (1,1); the fourth by symmetry. */
typedef double rkfd_qd4 __attribute__((ext_vector_type(4)));
RKFD_DEV void rkfd_vert_s_mfma(const double *W, int ldq, int n, int r, double *S, int ld)
{
  const int lane = LANE();
  const int kk = lane >> 4, cc = lane & 15;
  rkfd_qd4 c00 = { 0, 0, 0, 0 }, c01 = { 0, 0, 0, 0 }, c11 = { 0, 0, 0, 0 };
  for( int k0=0; k0<n; k0+=4 ){
    const int k = k0 + kk;
    const double a0 = ( k < n && cc < r ) ? W[k*ldq + cc] : 0.0;
    const double a1 = ( k < n && 16+cc < r ) ? W[k*ldq + 16 + cc] : 0.0;
    c00 = __builtin_amdgcn_mfma_f64_16x16x4f64( a0, a0, c00, 0, 0, 0 );
    if( r > 16 ){
      c01 = __builtin_amdgcn_mfma_f64_16x16x4f64( a0, a1, c01, 0, 0, 0 );
      c11 = __builtin_amdgcn_mfma_f64_16x16x4f64( a1, a1, c11, 0, 0, 0 );
    }
  }
#pragma unroll
  for( int rg=0; rg<4; rg++ ){
    const int row = kk + 4*rg;
    if( row < r && cc < r ) S[row*ld + cc] = c00[rg];
    if( row < r && 16+cc < r ){ S[row*ld + 16+cc] = c01[rg]; S[( 16+cc )*ld + row] = c01[rg]; }
    if( 16+row < r && 16+cc < r ) S[( 16+row )*ld + 16+cc] = c11[rg];
  }
}
#endif

#define RKFD_QP_ASM_TOL 1.0e-8
#define RKFD_QP_MAXITER 256

/* ------------------------------------------------------------------------ */
/* The QP of the Vert plugin:  min f'Qf/2 + c'f  s.t.  mu' f_n + sin(th_i) f_1 + cos(th_i) f_2 >= 0
 * for the P faces of every contact's friction pyramid, solved by the reference's active-set method
 * from the start point f_n = 1 (src/rkfd_opt_qp.c:43-181).  lane = constraint (P nc <= 64).
 *
 * The reference solves each KKT system [ -Q N' ; N 0 ] with a Moore-Penrose routine, because with
 * three or more faces of one pyramid active (a contact carrying no force sits at the apex with all
 * P faces active) N has dependent rows.  The structure makes that solve cheap and exact here: Q is
 * positive definite and rows of different contacts touch different unknowns, so
 *   - a contact with >= 3 active faces is the equality f_c = 0: its rows are replaced by the three
 *     unit rows, and the reduced constraint matrix C (<= 3 rows per contact) has full row rank;
 *   - with Q = LL' factored once per evaluation:  W = L^-1 C',  (W'W) lambda = W' L^-1 c,  f = L^-T ( W lambda - L^-1 c ),
 *     every solve by triangular substitution.  (Going through an explicit Q^-1 is 25 % faster but its
 *     rounding noise, ~cond(Q) eps, is enough to flip the method's 1e-12 knife-edge decisions: agreement
 *     with the oracle over 128 random box drops x 60 steps fell from 113 to 62 instances.)
 *   - the minimum-norm multipliers of the original rows are y = G (G'G)^-1 lambda_c per contact
 *     (G = its active rows), which is what the pseudo-inverse returns.
 * In:  L.MA = A (n x n, ld = n+1, without relaxation), L.MB = c_vel (bias incl. compensation).
 * Out: L.MF = f / dt; returns the mask of the constraints active at the solution (bit = lane). */
template<bool prof> RKFD_DEV unsigned long long rkfd_vert_qp(const rkfdDevModel &m, const rkfdLds &L, int nc, unsigned long long *pc)
{
  unsigned long long q0 = prof ? RKFD_CLOCK() : 0ull, q1;
#define VST(k) do{ if( prof ){ q1 = RKFD_CLOCK(); pc[k] += q1 - q0; q0 = q1; } }while(0)
  const int lane = LANE();
  const int n = 3*nc, ld = n+1, ldq = n;      /* ld: the matrix that held A (now S); ldq: Q's factor and W */
  const int P = m.pyramid, mc = P*nc;
  /* the factor in registers where it fits (see rkfdQpFactor; the host decides: m.vert_rigid == 2 <=> 3 maxrg <= RKFD_QP_NQ_MAX).
   * The LDS of such a world has no W block: W takes the storage of A once Q and c are formed, and the Schur complement S (packed)
   * that of the factor once the registers hold it. */
  const bool reg = RKFD_QP_NQ > 0 && m.vert_rigid == 2;
  double *Q = L.QL, *W = reg ? L.MA : L.QW;
  double *cv = L.QV, *zv = L.QV + n, *ans = L.QV + 2*n, *lam = L.QV + 3*n, *dv = L.QV + 4*n;
  double *xv = L.MB;                          /* the bias vector is dead once c = A'b is formed */
  const bool onc = lane < mc;                 /* this lane is a constraint */
  const int cc = onc ? lane/P : 0;            /* its contact */

  /* pyramid row of this lane (_rkFDSolverFrictionConstraint): ( mu cos(pi/P), sin(th), cos(th) ), th = 2 pi i/P - pi/P */
  double g0 = 0, g1 = 0, g2 = 0;
  {
    const double PI = 3.14159265358979323846;
    const int jc = L.lrg[cc], ci = RKFD_CI_CI( L.CIp[jc] );
    const double mu = L.typ[jc] == RKFD_KF ? m.ci_kf[ci] : m.ci_sf[ci];
    double s0, c0, s1, c1;
    /* d_sincos takes any argument; the reference evaluates sin/cos of th+offset with th accumulated by additions */
    double th = 0.0;
    for( int k=0; k<( onc ? lane - cc*P : 0 ); k++ ) th += 2.0*PI/P;
    d_sincos( th + ( -PI/P ), &s1, &c1 );
    d_sincos( 0.0 + ( -PI/P ), &s0, &c0 );
    if( onc ){ g0 = mu*c0; g1 = s1; g2 = c1; }
  }
  /* c = A'c first: with the factor in registers the Gram product below lands on A's own storage */
  if( lane < n ){
    double s = 0;
#pragma unroll 8
    for( int r=0; r<n; r++ ) s = fma( L.MA[r*ld+lane], L.MB[r], s );
    cv[lane] = s;
    ans[lane] = ( lane%3 == 0 ) ? 1.0 : 0.0;           /* _rkFDSolverQPASMInit */
  }
  SYNC();
  /* q = A'A + L (lower triangle, packed by rows) */
#ifndef RKFD_EMU
  if( ( m.mlcp_mfma & 4 ) && n <= 32 ){
    /* the Gram product on the matrix cores (v_mfma_f64_16x16x4_f64) into W's storage, free until the first iteration, then packed:
     * a full 24 x 24 x 24 product, where the VALU loop below loads 2 x 24 values from LDS per entry: q:setup 337 k -> 257 k cycles per
     * step, config 4 under the Vert plugin 3.30 -> 3.48 M steps/s (profiles/r02_vert_mfma_ab.txt) */
    rkfd_vert_s_mfma( L.MA, ld, n, n, W, ldq );
    SYNC();
    for( int t0=0; t0<n*n; t0+=RKFD_WAVE ){
      const int t = t0 + lane, i = t/n, k = t - i*n;
      if( t < n*n && k <= i ) Q[RKFD_QI( i, k )] = W[i*ldq+k] + ( i == k ? m.ci_l[RKFD_CI_CI( L.CIp[L.lrg[i/3]] )] : 0.0 );
    }
    SYNC();
  } else
#endif
  for( int t0=0; t0<n*n; t0+=RKFD_WAVE ){
    const int t = t0 + lane, i = t/n, k = t - i*n;
    if( t < n*n && k <= i ){
      double s = 0;
#pragma unroll 8
      for( int r=0; r<n; r++ ) s = fma( L.MA[r*ld+i], L.MA[r*ld+k], s );
      if( i == k ) s += m.ci_l[RKFD_CI_CI( L.CIp[L.lrg[i/3]] )];
      Q[RKFD_QI( i, k )] = s;
    }
  }
  SYNC();
  if( reg ) rkfd_qreg_chol<RKFD_QP_NQ>( Q, n ); else rkfd_w_chol<true>( Q, 0, n );
  rkfdQpFactor F;
  F.rd = 0.0;
  if( reg ){ SYNC(); rkfd_qreg_load( F, Q, n ); }
  /* z = L^-1 c */
  {
    const double ci = lane < n ? cv[lane] : 0.0;
    const double zi = reg ? rkfd_qreg_fwd( F, n, ci ) : rkfd_w_fwd<true>( Q, 0, n, ci );
    if( lane < n ) zv[lane] = zi;
  }
  SYNC();
  VST(24);
  /* initial active set (_rkFDQPSolveASMInitIndex) */
  int act = 0;
  if( onc ){
    const double cnd = g0*ans[3*cc] + g1*ans[3*cc+1] + g2*ans[3*cc+2];
    act = fabs( cnd - 0.0 ) < RKFD_DEV_TOL;
  }
  unsigned long long hmask = 0; double hobj = 0; int nhist = 0;      /* lane h keeps visited basis h */
  unsigned long long mask = 0;
  int fail = 0;
  for( int iter=0; ; iter++ ){
    if( iter >= RKFD_QP_MAXITER ){ fail = 1; break; }
    mask = BALLOT( act );
    /* reduced, full-rank constraint rows: per contact the active faces themselves (1 or 2), or the three unit rows (>= 3) */
    const unsigned long long cmask = ( P >= 64 ? ~0ull : ( ( 1ull << P ) - 1ull ) );
    int roff = 0, kc = 0, rho = 0;
    int r = 0;
    for( int c=0; c<nc; c++ ){
      const int k = __builtin_popcountll( ( mask >> ( c*P ) ) & cmask );
      if( c < cc ) roff += k < 3 ? k : 3;
      if( c == cc ) kc = k;
      r += k < 3 ? k : 3;
    }
    if( onc ) rho = __builtin_popcountll( ( mask >> ( cc*P ) ) & cmask & ( ( 1ull << ( lane - cc*P ) ) - 1ull ) );
    if( onc && act ){
      if( kc < 3 ){
        L.CR[3*( roff+rho )] = g0; L.CR[3*( roff+rho )+1] = g1; L.CR[3*( roff+rho )+2] = g2; L.CRC[roff+rho] = cc;
      } else if( rho < 3 ){
        L.CR[3*( roff+rho )] = rho == 0 ? 1.0 : 0.0; L.CR[3*( roff+rho )+1] = rho == 1 ? 1.0 : 0.0; L.CR[3*( roff+rho )+2] = rho == 2 ? 1.0 : 0.0;
        L.CRC[roff+rho] = cc;
      }
    }
    SYNC();
    /* W = L^-1 C' by forward substitution (lane = reduced row), S = W'W, rhs = W'z */
    if( reg ){
      /* lane = row, two columns per pass */
      for( int a0=0; a0<r; a0+=2 ){
        const int a1 = a0+1 < r ? a0+1 : a0;
        const int c30 = 3*L.CRC[a0], c31 = 3*L.CRC[a1];
        double y0 = lane == c30 ? L.CR[3*a0] : ( lane == c30+1 ? L.CR[3*a0+1] : ( lane == c30+2 ? L.CR[3*a0+2] : 0.0 ) );
        double y1 = lane == c31 ? L.CR[3*a1] : ( lane == c31+1 ? L.CR[3*a1+1] : ( lane == c31+2 ? L.CR[3*a1+2] : 0.0 ) );
        rkfd_qreg_fwd2( F, n, c30 < c31 ? c30 : c31, y0, y1 );
        if( lane < n ){
          W[lane*ldq+a0] = lane < c30 ? 0.0 : y0;
          if( a1 != a0 ) W[lane*ldq+a1] = lane < c31 ? 0.0 : y1;
        }
      }
    } else if( lane < r ){
      const int c3 = 3*L.CRC[lane];
      const double h0 = L.CR[3*lane], h1 = L.CR[3*lane+1], h2 = L.CR[3*lane+2];
      for( int i=0; i<n; i++ ){
        double sacc = i == c3 ? h0 : ( i == c3+1 ? h1 : ( i == c3+2 ? h2 : 0.0 ) );
#pragma unroll 8
        for( int j=c3; j<i; j++ ) sacc -= Q[RKFD_QI( i, j )]*W[j*ldq+lane];
        W[i*ldq+lane] = i < c3 ? 0.0 : sacc*Q[RKFD_QI( i, i )];
      }
    }
    SYNC();
    VST(25);
    double *S = reg ? Q : L.MA;      /* (reg: packed lower triangle) */
    /* (on the matrix cores this product gained nothing: 262 k -> 271 k cycles per step, profiles/r02_vert_mfma_ab.txt - with 2 .. 16
     * of 24 rows active it is a few passes of latency either way; the switch was removed in round 3) */
    for( int t0=0; t0<( r*( r+1 ) >> 1 ); t0+=RKFD_WAVE ){
      /* lane = entry ( a, b <= a ) of the lower triangle, counted row by row: every lane of a pass has work */
      const int t = t0 + lane;
      int a = (int)( ( sqrt( 8.0*t + 1.0 ) - 1.0 )*0.5 );
      if( ( a*( a+1 ) >> 1 ) > t ) a--;
      if( ( ( a+1 )*( a+2 ) >> 1 ) <= t ) a++;
      const int b = t - ( a*( a+1 ) >> 1 );
      if( t < ( r*( r+1 ) >> 1 ) ){
        /* column a of W = L^-1 C' is zero above the first unknown of its contact (the forward substitution starts there):
         * the skipped terms are exact zeros, the sum is bit for bit the same */
        const int ca = L.CRC[a], cb = L.CRC[b];
        double sacc = 0;
#pragma unroll 8
        for( int i=3*( ca > cb ? ca : cb ); i<n; i++ ) sacc = fma( W[i*ldq+a], W[i*ldq+b], sacc );
        if( reg ) S[RKFD_QI( a, b )] = sacc; else { S[a*ld+b] = sacc; S[b*ld+a] = sacc; }
      }
    }
    double rl = 0;
    if( lane < r ){
#pragma unroll 8
      for( int i=3*L.CRC[lane]; i<n; i++ ) rl = fma( W[i*ldq+lane], zv[i], rl );
    }
    SYNC();
    VST(26);
    if( reg ){
      rkfd_w_chol<true>( S, 0, r );
      const double y = rkfd_w_fwd<true>( S, 0, r, rl );
      const double l = rkfd_w_back<true>( S, 0, r, y );
      if( lane < r ) lam[lane] = l;
    } else {
      rkfd_w_chol<false>( S, ld, r );
      const double y = rkfd_w_fwd<false>( S, ld, r, rl );
      const double l = rkfd_w_back<false>( S, ld, r, y );
      if( lane < r ) lam[lane] = l;
    }
    SYNC();
    VST(27);
    /* f = L^-T ( W lambda - z ) */
    {
      double ti = 0;
      if( lane < n ){
#pragma unroll 8
        for( int a=0; a<r; a++ ) ti = fma( W[lane*ldq+a], lam[a], ti );
        ti -= zv[lane];
      }
      const double xi = reg ? rkfd_qreg_back( F, n, ti ) : rkfd_w_back<true>( Q, 0, n, ti );
      if( lane < n ) xv[lane] = xi;
    }
    SYNC();
    VST(28);
    const bool moved = BALLOT( lane < n && !( fabs( xv[lane] - ans[lane] ) < RKFD_DEV_TOL ) ) != 0ull;
#if defined(RKFD_EMU) && defined(RKFD_QP_TRACE)
    if( lane == 0 ){ printf( "dev it %d mask %016llx r %d moved %d x", iter, mask, r, (int)moved ); for( int i=0; i<n; i++ ) printf( " %.6e", xv[i] ); printf( "\n" ); }
#endif
    if( !moved ){
      if( lane < n ) ans[lane] = xv[lane];
      /* multipliers of the original rows */
      double y = 0;
      if( onc && act ){
        if( kc < 3 ) y = lam[roff+rho];
        else {
          /* y = g . (G'G)^-1 lambda_c over the active faces of this contact */
          double a00 = 0, a01 = 0, a02 = 0, a11 = 0, a12 = 0, a22 = 0;
          const double PI = 3.14159265358979323846;
          double th = 0.0;
          for( int k=0; k<P; k++, th+=2.0*PI/P ){
            if( !( ( mask >> ( cc*P+k ) ) & 1ull ) ) continue;
            double sk, ck;
            d_sincos( th + ( -PI/P ), &sk, &ck );
            a00 += g0*g0; a01 += g0*sk; a02 += g0*ck; a11 += sk*sk; a12 += sk*ck; a22 += ck*ck;
          }
          const double l0 = lam[roff], l1 = lam[roff+1], l2 = lam[roff+2];
          const double c00 = a11*a22 - a12*a12, c01 = a02*a12 - a01*a22, c02 = a01*a12 - a02*a11;
          const double c11 = a00*a22 - a02*a02, c12 = a01*a02 - a00*a12, c22 = a00*a11 - a01*a01;
          const double det = a00*c00 + a01*c01 + a02*c02;
          const double u0 = ( c00*l0 + c01*l1 + c02*l2 )/det, u1 = ( c01*l0 + c11*l1 + c12*l2 )/det, u2 = ( c02*l0 + c12*l1 + c22*l2 )/det;
          y = g0*u0 + g1*u1 + g2*u2;
        }
      }
      SYNC();
      if( BALLOT( onc && act && y < 0 ) == 0ull ) break;                 /* found the optimal solution */
      const double ymin = WMIN( ( onc && act ) ? y : HUGE_VAL );
      if( onc && act && fabs( y - ymin ) < RKFD_QP_ASM_TOL ) act = 0;
      VST(29);
      continue;
    }
    /* STEP2: towards the equality-constrained minimiser as far as the inactive constraints allow */
    if( lane < n ) dv[lane] = xv[lane] - ans[lane];
    SYNC();
    double tq = HUGE_VAL;
    if( onc && !act ){
      const double gd = g0*dv[3*cc] + g1*dv[3*cc+1] + g2*dv[3*cc+2];
      if( gd < 0 ) tq = ( 0.0 - ( g0*ans[3*cc] + g1*ans[3*cc+1] + g2*ans[3*cc+2] ) )/gd;
    }
    double tmin = WMIN( tq );
    if( !( tmin < 1.0 ) ) tmin = 1.0;
#if defined(RKFD_EMU) && defined(RKFD_QP_TRACE)
    if( lane == 0 ) printf( "dev    step t %.12e\n", tmin );
#endif
    if( lane < n ) ans[lane] += tmin*dv[lane];
    SYNC();
    if( onc && !act && fabs( g0*ans[3*cc] + g1*ans[3*cc+1] + g2*ans[3*cc+2] - 0.0 ) < RKFD_DEV_TOL ) act = 1;
    /* circulation check (degeneracy): same basis seen before with the same objective value.
     * f'Qf/2 = |L'f|^2/2 with the stored factor (diagonal kept as 1/L_ii) */
    double part = 0;
    if( reg ){
      const double ai = lane < n ? ans[lane] : 0.0;
      const double u = rkfd_qreg_ltv( F, n, ai );
      if( lane < n ) part = 0.5*u*u + cv[lane]*ai;
    } else if( lane < n ){
      double u = ans[lane]/Q[RKFD_QI( lane, lane )];
#pragma unroll 8
      for( int j=lane+1; j<n; j++ ) u = fma( Q[RKFD_QI( j, lane )], ans[j], u );
      part = 0.5*u*u + cv[lane]*ans[lane];
    }
    /* (the sum only feeds the circulation check's comparison to 1e-8: its association is free) */
    const double objv = WSUM( part );
    const unsigned long long nmask = BALLOT( act );
    const bool seen = lane < nhist && hmask == nmask && !( fabs( hobj/objv - 1.0 ) > RKFD_QP_ASM_TOL );
    if( BALLOT( seen ) != 0ull ){ mask = nmask; break; }
    if( nhist >= RKFD_WAVE ){ fail = 1; break; }
    if( lane == nhist ){ hmask = nmask; hobj = objv; }
    nhist++;
    VST(30);
  }
  mask = BALLOT( act );
  /* a Q that is not numerically positive definite (no relaxation and a singular A) ends in NaNs: report it */
  if( BALLOT( lane < n && !( ans[lane] == ans[lane] ) ) != 0ull ) fail = 1;
  if( fail && lane == 0 ) L.cnt[CNT_QPF] = 1;
  SYNC();
  if( lane < n ) L.MF[lane] = ans[lane]/m.dt;
  SYNC();
#undef VST
  return mask;
}


/* ------------------------------------------------------------------------ */
/* The WIDE form of the same QP: more unknowns or more pyramid faces than the wavefront has lanes (3 Nc > 64 or P Nc > 64; the
 * reference has no such limit, src/rkfd_vert.c:73-103 - config 5's 24 contact vertices under the default plugin are 72 unknowns
 * and 192 faces).  The method, its start point, its tolerances and the order of its decisions are those of rkfd_vert_qp above;
 * what changes is where things live: every vector of the solver is in LDS, every loop over unknowns / faces / reduced rows is
 * strided by the wavefront, the active set is a byte per face (and three 64-bit words where bases are compared), and the dense
 * routines take their pivots from LDS instead of v_readlane.  Built for correctness, not speed: such a world holds one
 * instance per CU (its matrices are 100 KB of LDS). */
#define RKFD_WI(r,c) ( pk ? ( ( (r)*( (r)+1 ) ) >> 1 ) + (c) : (r)*ld + (c) )
template<bool pk> RKFD_DEV void rkfd_ww_chol(double *Mx, int ld, int n, double *tmp)
{
  const int lane = LANE();
  for( int k=0; k<n; k++ ){
    for( int i=lane; i<n; i+=RKFD_WAVE ) if( i >= k ){
      double s = Mx[RKFD_WI( i, k )];
      for( int j=0; j<k; j++ ) s -= Mx[RKFD_WI( i, j )]*Mx[RKFD_WI( k, j )];
      tmp[i] = s;
    }
    SYNC();
    const double rinv = RKFD_RCP( sqrt( tmp[k] ) );
    for( int i=lane; i<n; i+=RKFD_WAVE ) if( i >= k ) Mx[RKFD_WI( i, k )] = i == k ? rinv : tmp[i]*rinv;
    SYNC();
  }
}
/* v <- L^-1 v, v in LDS */
template<bool pk> RKFD_DEV void rkfd_ww_fwd(const double *Mx, int ld, int n, double *v)
{
  const int lane = LANE();
  for( int j=0; j<n; j++ ){
    const double yj = v[j]*Mx[RKFD_WI( j, j )];
    SYNC();
    for( int i=lane; i<n; i+=RKFD_WAVE ){
      if( i == j ) v[i] = yj;
      else if( i > j ) v[i] -= Mx[RKFD_WI( i, j )]*yj;
    }
    SYNC();
  }
}
/* v <- L^-T v */
template<bool pk> RKFD_DEV void rkfd_ww_back(const double *Mx, int ld, int n, double *v)
{
  const int lane = LANE();
  for( int j=n-1; j>=0; j-- ){
    const double xj = v[j]*Mx[RKFD_WI( j, j )];
    SYNC();
    for( int i=lane; i<n; i+=RKFD_WAVE ){
      if( i == j ) v[i] = xj;
      else if( i < j ) v[i] -= Mx[RKFD_WI( j, i )]*xj;
    }
    SYNC();
  }
}
#undef RKFD_WI

/* In / out as rkfd_vert_qp; the active flags stay in L.QA[face] for the caller's stick / slip decision. */
template<bool prof> RKFD_DEV void rkfd_vert_qp_wide(const rkfdDevModel &m, const rkfdLds &L, int nc, unsigned long long *pc)
{
  unsigned long long q0 = prof ? RKFD_CLOCK() : 0ull, q1;
#define VST(k) do{ if( prof ){ q1 = RKFD_CLOCK(); pc[k] += q1 - q0; q0 = q1; } }while(0)
  const int lane = LANE();
  const int n = 3*nc, ld = n+1, ldq = n;
  const int P = m.pyramid, mc = P*nc;
  double *Q = L.QL, *W = L.QW, *S = L.MA;
  double *cv = L.QV, *zv = L.QV + n, *ans = L.QV + 2*n, *lam = L.QV + 3*n, *dv = L.QV + 4*n;
  double *xv = L.MB;
  double *G = L.QG, *yv = L.QY;
  unsigned char *act = L.QA, *kcn = L.QA + P*m.maxrg;
  const double PI = 3.14159265358979323846;

  /* pyramid rows (_rkFDSolverFrictionConstraint), face q = contact q / P, direction q % P */
  {
    double s0, c0;
    d_sincos( 0.0 + ( -PI/P ), &s0, &c0 );
    for( int q=lane; q<mc; q+=RKFD_WAVE ){
      const int cc = q/P, kf = q - cc*P;
      const int jc = L.lrg[cc], ci = RKFD_CI_CI( L.CIp[jc] );
      const double mu = L.typ[jc] == RKFD_KF ? m.ci_kf[ci] : m.ci_sf[ci];
      double th = 0.0, s1, c1;
      for( int k=0; k<kf; k++ ) th += 2.0*PI/P;
      d_sincos( th + ( -PI/P ), &s1, &c1 );
      G[3*q] = mu*c0; G[3*q+1] = s1; G[3*q+2] = c1;
    }
  }
  /* c = A'c, q = A'A + L (packed lower triangle), the start point */
  for( int i=lane; i<n; i+=RKFD_WAVE ){
    double s = 0;
    for( int r=0; r<n; r++ ) s = fma( L.MA[r*ld+i], L.MB[r], s );
    cv[i] = s;
    ans[i] = ( i%3 == 0 ) ? 1.0 : 0.0;
  }
  for( int t=lane; t<n*n; t+=RKFD_WAVE ){
    const int i = t/n, k = t - i*n;
    if( k <= i ){
      double s = 0;
      for( int r=0; r<n; r++ ) s = fma( L.MA[r*ld+i], L.MA[r*ld+k], s );
      if( i == k ) s += m.ci_l[RKFD_CI_CI( L.CIp[L.lrg[i/3]] )];
      Q[RKFD_QI( i, k )] = s;
    }
  }
  SYNC();
  rkfd_ww_chol<true>( Q, 0, n, dv );
  for( int i=lane; i<n; i+=RKFD_WAVE ) zv[i] = cv[i];
  SYNC();
  rkfd_ww_fwd<true>( Q, 0, n, zv );
  VST(24);
  /* initial active set */
  for( int q=lane; q<mc; q+=RKFD_WAVE ){
    const int c3 = 3*( q/P );
    act[q] = fabs( G[3*q]*ans[c3] + G[3*q+1]*ans[c3+1] + G[3*q+2]*ans[c3+2] - 0.0 ) < RKFD_DEV_TOL;
  }
  SYNC();
  unsigned long long hw0 = 0, hw1 = 0, hw2 = 0; double hobj = 0; int nhist = 0;      /* lane h keeps visited basis h */
  int fail = 0;
  for( int iter=0; ; iter++ ){
    if( iter >= RKFD_QP_MAXITER ){ fail = 1; break; }
    /* reduced, full-rank constraint rows: per contact its active faces (1 or 2) or the three unit rows (>= 3) */
    for( int c=lane; c<nc; c+=RKFD_WAVE ){
      int k = 0;
      for( int f=0; f<P; f++ ) k += act[c*P+f];
      kcn[c] = (unsigned char)k;
    }
    SYNC();
    int r = 0;
    for( int c=0; c<nc; c++ ) r += kcn[c] < 3 ? kcn[c] : 3;
    for( int q=lane; q<mc; q+=RKFD_WAVE ) if( act[q] ){
      const int cc = q/P, kf = q - cc*P, kc = kcn[cc];
      int roff = 0, rho = 0;
      for( int c=0; c<cc; c++ ) roff += kcn[c] < 3 ? kcn[c] : 3;
      for( int f=0; f<kf; f++ ) rho += act[cc*P+f];
      if( kc < 3 ){
        L.CR[3*( roff+rho )] = G[3*q]; L.CR[3*( roff+rho )+1] = G[3*q+1]; L.CR[3*( roff+rho )+2] = G[3*q+2]; L.CRC[roff+rho] = (unsigned char)cc;
      } else if( rho < 3 ){
        L.CR[3*( roff+rho )] = rho == 0 ? 1.0 : 0.0; L.CR[3*( roff+rho )+1] = rho == 1 ? 1.0 : 0.0; L.CR[3*( roff+rho )+2] = rho == 2 ? 1.0 : 0.0;
        L.CRC[roff+rho] = (unsigned char)cc;
      }
    }
    SYNC();
    /* W = L^-1 C' (lane = reduced row), S = W'W, rhs = W'z */
    for( int a=lane; a<r; a+=RKFD_WAVE ){
      const int c3 = 3*L.CRC[a];
      const double h0 = L.CR[3*a], h1 = L.CR[3*a+1], h2 = L.CR[3*a+2];
      for( int i=0; i<n; i++ ){
        double sacc = i == c3 ? h0 : ( i == c3+1 ? h1 : ( i == c3+2 ? h2 : 0.0 ) );
        for( int j=c3; j<i; j++ ) sacc -= Q[RKFD_QI( i, j )]*W[j*ldq+a];
        W[i*ldq+a] = i < c3 ? 0.0 : sacc*Q[RKFD_QI( i, i )];
      }
    }
    SYNC();
    VST(25);
    for( int t=lane; t<( r*( r+1 ) >> 1 ); t+=RKFD_WAVE ){
      int a = (int)( ( sqrt( 8.0*t + 1.0 ) - 1.0 )*0.5 );
      if( ( a*( a+1 ) >> 1 ) > t ) a--;
      if( ( ( a+1 )*( a+2 ) >> 1 ) <= t ) a++;
      const int b = t - ( a*( a+1 ) >> 1 );
      const int ca = L.CRC[a], cb = L.CRC[b];
      double sacc = 0;
      for( int i=3*( ca > cb ? ca : cb ); i<n; i++ ) sacc = fma( W[i*ldq+a], W[i*ldq+b], sacc );
      S[a*ld+b] = sacc; S[b*ld+a] = sacc;
    }
    for( int a=lane; a<r; a+=RKFD_WAVE ){
      double rl = 0;
      for( int i=3*L.CRC[a]; i<n; i++ ) rl = fma( W[i*ldq+a], zv[i], rl );
      lam[a] = rl;
    }
    SYNC();
    VST(26);
    rkfd_ww_chol<false>( S, ld, r, dv );
    rkfd_ww_fwd<false>( S, ld, r, lam );
    rkfd_ww_back<false>( S, ld, r, lam );
    VST(27);
    /* f = L^-T ( W lambda - z ) */
    for( int i=lane; i<n; i+=RKFD_WAVE ){
      double ti = 0;
      for( int a=0; a<r; a++ ) ti = fma( W[i*ldq+a], lam[a], ti );
      xv[i] = ti - zv[i];
    }
    SYNC();
    rkfd_ww_back<true>( Q, 0, n, xv );
    VST(28);
    bool mv = false;
    for( int i=lane; i<n; i+=RKFD_WAVE ) mv = mv || !( fabs( xv[i] - ans[i] ) < RKFD_DEV_TOL );
    if( !ANY( mv ) ){
      SYNC();
      for( int i=lane; i<n; i+=RKFD_WAVE ) ans[i] = xv[i];
      /* multipliers of the original rows */
      bool neg = false;
      double ylo = HUGE_VAL;
      for( int q=lane; q<mc; q+=RKFD_WAVE ) if( act[q] ){
        const int cc = q/P, kf = q - cc*P, kc = kcn[cc];
        int roff = 0, rho = 0;
        for( int c=0; c<cc; c++ ) roff += kcn[c] < 3 ? kcn[c] : 3;
        for( int f=0; f<kf; f++ ) rho += act[cc*P+f];
        double y;
        if( kc < 3 ) y = lam[roff+rho];
        else {
          /* y = g . (G'G)^-1 lambda_c over the active faces of this contact */
          const double g0 = G[3*q];
          double a00 = 0, a01 = 0, a02 = 0, a11 = 0, a12 = 0, a22 = 0;
          double th = 0.0;
          for( int k=0; k<P; k++, th+=2.0*PI/P ){
            if( !act[cc*P+k] ) continue;
            double sk, ck;
            d_sincos( th + ( -PI/P ), &sk, &ck );
            a00 += g0*g0; a01 += g0*sk; a02 += g0*ck; a11 += sk*sk; a12 += sk*ck; a22 += ck*ck;
          }
          const double l0 = lam[roff], l1 = lam[roff+1], l2 = lam[roff+2];
          const double c00 = a11*a22 - a12*a12, c01 = a02*a12 - a01*a22, c02 = a01*a12 - a02*a11;
          const double c11 = a00*a22 - a02*a02, c12 = a01*a02 - a00*a12, c22 = a00*a11 - a01*a01;
          const double det = a00*c00 + a01*c01 + a02*c02;
          const double u0 = ( c00*l0 + c01*l1 + c02*l2 )/det, u1 = ( c01*l0 + c11*l1 + c12*l2 )/det, u2 = ( c02*l0 + c12*l1 + c22*l2 )/det;
          y = g0*u0 + G[3*q+1]*u1 + G[3*q+2]*u2;
        }
        yv[q] = y;
        neg = neg || y < 0;
        ylo = y < ylo ? y : ylo;
      }
      SYNC();
      if( !ANY( neg ) ) break;                 /* found the optimal solution */
      const double ymin = WMIN( ylo );
      for( int q=lane; q<mc; q+=RKFD_WAVE ) if( act[q] && fabs( yv[q] - ymin ) < RKFD_QP_ASM_TOL ) act[q] = 0;
      SYNC();
      VST(29);
      continue;
    }
    /* STEP2: towards the equality-constrained minimiser as far as the inactive constraints allow */
    for( int i=lane; i<n; i+=RKFD_WAVE ) dv[i] = xv[i] - ans[i];
    SYNC();
    double tq = HUGE_VAL;
    for( int q=lane; q<mc; q+=RKFD_WAVE ) if( !act[q] ){
      const int c3 = 3*( q/P );
      const double gd = G[3*q]*dv[c3] + G[3*q+1]*dv[c3+1] + G[3*q+2]*dv[c3+2];
      if( gd < 0 ){
        const double t = ( 0.0 - ( G[3*q]*ans[c3] + G[3*q+1]*ans[c3+1] + G[3*q+2]*ans[c3+2] ) )/gd;
        tq = t < tq ? t : tq;
      }
    }
    double tmin = WMIN( tq );
    if( !( tmin < 1.0 ) ) tmin = 1.0;
    SYNC();
    for( int i=lane; i<n; i+=RKFD_WAVE ) ans[i] += tmin*dv[i];
    SYNC();
    for( int q=lane; q<mc; q+=RKFD_WAVE ) if( !act[q] ){
      const int c3 = 3*( q/P );
      if( fabs( G[3*q]*ans[c3] + G[3*q+1]*ans[c3+1] + G[3*q+2]*ans[c3+2] - 0.0 ) < RKFD_DEV_TOL ) act[q] = 1;
    }
    /* circulation check (degeneracy): same basis seen before with the same objective value; f'Qf/2 = |L'f|^2/2 */
    double part = 0;
    for( int i=lane; i<n; i+=RKFD_WAVE ){
      double u = ans[i]/Q[RKFD_QI( i, i )];
      for( int j=i+1; j<n; j++ ) u = fma( Q[RKFD_QI( j, i )], ans[j], u );
      part += 0.5*u*u + cv[i]*ans[i];
    }
    const double objv = WSUM( part );
    SYNC();
    const unsigned long long w0 = BALLOT( lane < mc && act[lane < mc ? lane : 0] );
    const unsigned long long w1 = BALLOT( 64+lane < mc && act[64+lane < mc ? 64+lane : 0] );
    const unsigned long long w2 = BALLOT( 128+lane < mc && act[128+lane < mc ? 128+lane : 0] );
    const bool seen = lane < nhist && hw0 == w0 && hw1 == w1 && hw2 == w2 && !( fabs( hobj/objv - 1.0 ) > RKFD_QP_ASM_TOL );
    if( ANY( seen ) ) break;
    if( nhist >= RKFD_WAVE ){ fail = 1; break; }
    if( lane == nhist ){ hw0 = w0; hw1 = w1; hw2 = w2; hobj = objv; }
    nhist++;
    VST(30);
  }
  SYNC();
  bool bad = false;
  for( int i=lane; i<n; i+=RKFD_WAVE ) bad = bad || !( ans[i] == ans[i] );
  if( ANY( bad ) ) fail = 1;
  if( fail && lane == 0 ) L.cnt[CNT_QPF] = 1;
  SYNC();
  for( int i=lane; i<n; i+=RKFD_WAVE ) L.MF[i] = ans[i]/m.dt;
  SYNC();
#undef VST
}

#endif /* RKFD_DEV_VERTQP_H */
