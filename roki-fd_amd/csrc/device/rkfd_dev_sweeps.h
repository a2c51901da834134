/* rkfd_dev_sweeps.h - part of the device code of the batched rkFDUpdate step (see rkfd_device.h):
 * 6x6 Cholesky helpers, the host-built sweep schedule, ABA sweep 2 (leaf to root) and sweep 3 (root to leaf, plain and delta).
 * Included by rkfd_device.h only, in this order; compiles for gfx950 and under the lane emulator. */
#ifndef RKFD_DEV_SWEEPS_H
#define RKFD_DEV_SWEEPS_H

/* ------------------------------------------------------------------------ */
/* in-place Cholesky of the 6x6 at A (row-major, lower part used), one lane.  The diagonal
 * stores 1/L_jj so that the factorisation and the solves multiply instead of dividing. */
/* the factor of a float joint's articulated inertia: lower triangle, packed by rows (21 doubles) */
#define RKFD_TRI(i,k) ( (i)*( (i)+1 )/2 + (k) )
RKFD_DEV void d_chol6_inplace(double *A)
{
  /* the lower triangle is pulled into registers in one batch of loads, factored there and written back */
  double a[6][6];
#pragma unroll
  for( int i=0; i<6; i++ )
#pragma unroll
    for( int k=0; k<6; k++ ) if( k <= i ) a[i][k] = A[RKFD_TRI( i, k )];
#pragma unroll
  for( int j=0; j<6; j++ ){
    double s = a[j][j];
#pragma unroll
    for( int k=0; k<6; k++ ) if( k < j ) s -= a[j][k]*a[j][k];
    const double inv = RKFD_RCP( sqrt( s ) );
    a[j][j] = inv;
#pragma unroll
    for( int i=0; i<6; i++ ) if( i > j ){
      double t = a[i][j];
#pragma unroll
      for( int k=0; k<6; k++ ) if( k < j ) t -= a[i][k]*a[j][k];
      a[i][j] = t*inv;
    }
  }
#pragma unroll
  for( int i=0; i<6; i++ )
#pragma unroll
    for( int k=0; k<6; k++ ) if( k <= i ) A[RKFD_TRI( i, k )] = a[i][k];
}
/* the factor is pulled into registers in one batch of loads before a substitution starts: left to itself the compiler
 * sinks each load to its first use, and a 6x6 solve becomes eleven LDS round trips in a row */
RKFD_DEV void d_chol6_load(const double *Lm, double *Lr)
{
#pragma unroll
  for( int k=0; k<21; k++ ) Lr[k] = Lm[k];
  RKFD_SCHED_BARRIER();
}
/* forward substitution y = L^-1 b and back substitution x = L^-T y with that factor (in registers) */
RKFD_DEV void d_chol6_fwd(const double *Lm, const double *b, double *y)
{
#pragma unroll
  for( int i=0; i<6; i++ ){
    double s = b[i];
#pragma unroll
    for( int k=0; k<6; k++ ) if( k < i ) s -= Lm[RKFD_TRI( i, k )]*y[k];
    y[i] = s*Lm[RKFD_TRI( i, i )];
  }
}
RKFD_DEV void d_chol6_back(const double *Lm, const double *y, double *x)
{
#pragma unroll
  for( int i=5; i>=0; i-- ){
    double s = y[i];
#pragma unroll
    for( int k=0; k<6; k++ ) if( k > i ) s -= Lm[RKFD_TRI( k, i )]*x[k];
    x[i] = s*Lm[RKFD_TRI( i, i )];
  }
}

/* ------------------------------------------------------------------------ */
/* one schedule record: what one 8-lane group does in one sweep iteration (packed by the host:
 * link, packed link info, nchild | flags<<8 | pool slot<<16 | float slot<<24 (slots +1, 0 = none),
 * offset of the children list) */
typedef struct { int i, li, w, coff; } rkfdRec;
#define REC_NCHILD(r) ( (r).w & 0xFF )
#define REC_FLAGS(r)  ( ( (r).w >> 8 ) & 0xFF )
#define REC_POOL(r)   ( ( ( (r).w >> 16 ) & 0xFF ) - 1 )
#define REC_FSLOT(r)  ( ( ( (r).w >> 24 ) & 0xFF ) - 1 )
RKFD_DEV rkfdRec rkfd_rec_load(const rkfdDevModel &m, int t, int g)
{
  /* t in [-2, nsched+1]: the schedule is padded with two empty iterations on both sides */
  rkfdRec r;
  /* (two instances per wavefront: four lane groups per iteration, the schedule built for that) */
  const int *p = m.sched + ( (size_t)( t+2 )*( 8/RKFD_W ) + g )*4;
  r.i = p[0]; r.li = p[1]; r.w = p[2]; r.coff = p[3];
  return r;
}

/* per-lane operands of one sweep-2 iteration.
 * row = row rr of the link's own spatial inertia about the world origin,
 *   [ A   m [r]x ;  m [r]x'   m 1 ],   A = Iw + m( |r|^2 1 - r r' ):
 * every entry is one of the 14 staged doubles (A sym, +m r, -m r, m, 0), so a row is six loads at
 * lane-constant offsets ro[] (2.6x less LDS than staging the 6x6, no arithmetic). */
typedef struct { double row[6], S[6], c[6], S_r, pb, tau, jm; } rkfdPre2;
RKFD_DEV void rkfd_row_offsets(int rr, int *ro)
{
  /* [r]x = [ 0 -z y ; z 0 -x ; -y x 0 ];  +m r at 6..8, -m r at 9..11, m at 12, 0 at 13 */
  const int t[6][6] = { { 0, 1, 2, 13, 11, 7 }, { 1, 3, 4, 8, 13, 9 }, { 2, 4, 5, 10, 6, 13 },
                        { 13, 8, 10, 12, 13, 13 }, { 11, 13, 6, 13, 12, 13 }, { 7, 9, 13, 13, 13, 12 } };
#pragma unroll
  for( int k=0; k<6; k++ ){
    int v = t[0][k];
#pragma unroll
    for( int q=1; q<6; q++ ) v = rr == q ? t[q][k] : v;
    ro[k] = v;
  }
}
RKFD_DEV void rkfd_pre2_load(const rkfdDevModel &m, const rkfdLds &L, int i, int rr, const int *ro, rkfdPre2 &p)
{
#pragma unroll
  for( int k=0; k<6; k++ ){ p.S[k] = L.S[6*i+k]; p.c[k] = L.C[6*i+k]; }
  p.S_r = L.S[6*i+rr];
  p.pb = L.PB[6*i+rr];
  p.tau = L.MS[3*i+2]; p.jm = m.mot_inertia[i];
#pragma unroll
  for( int k=0; k<6; k++ ) p.row[k] = L.IST[14*i+ro[k]];
}

/* ABA sweep 2 (leaf to root), level-synchronous; 8 lanes per link, lane r = row r of the 6x6:
 * articulated inertia and bias force (backward part of rkChainUpdateABI).
 * Software-pipelined: schedule records are fetched two iterations ahead, and along chains the
 * child's (Ia row, pa) stay in registers (schedule flag bit 0), so the dependent path of an
 * iteration is one LDS round trip + ALU + DPP + one swizzle. */
template<bool prof> RKFD_DEV void rkfd_phase_sweep2(const rkfdDevModel &m, const rkfdLds &L, unsigned long long *pc)
{
  const int lane = LANE();
  const int g = lane >> 3, r = lane & 7;
  const int rr = r < 6 ? r : 0;
  const int T = m.nsched;
  rkfdRec rec1 = rkfd_rec_load( m, T-1, g ), rec2 = rkfd_rec_load( m, T-2, g );
  int ro[6];
  rkfd_row_offsets( rr, ro );
  double crow[6] = {0,0,0,0,0,0}, cpa = 0;
  for( int t=T-1; t>=0; t-- ){
    unsigned long long q0 = 0, q1;
#define QST(k) do{ if( prof ){ q1 = RKFD_CLOCK(); pc[k] += q1 - q0; q0 = q1; } }while(0)
    if( prof ) q0 = RKFD_CLOCK();
    const rkfdRec rec = rec1;
    rec1 = rec2;
    rec2 = rkfd_rec_load( m, t-2, g );
    /* operands of this iteration (the other waves of the SIMD cover the LDS latency;
     * a second, prefetched operand set would cost ~60 VGPRs) */
    rkfdPre2 pre;
    rkfd_pre2_load( m, L, rec.i >= 0 ? rec.i : 0, rr, ro, pre );
    QST(8);
    const bool onl = rec.i >= 0;
    const bool on = onl && r < 6;
    const int i = onl ? rec.i : 0;
    /* (worlds with breakable float joints: what such a joint is in this evaluation stands in the link info in LDS, rkfd_dev_brf.h) */
    const int jt = onl ? RKFD_LI_JT( m.has_brf ? L.LI[i] : rec.li ) : RKFD_JOINT_FIXED;
    const bool is1 = RKFD_JT_IS1( jt );
    const bool isf = jt == RKFD_JOINT_FLOAT;
    const bool isb = m.has_brf && onl && L.BRK[i] == RKFD_BRF_ATTACHED;      /* an attached breakable joint: a fixed joint today */
    double row[6], pr = pre.pb;
#pragma unroll
    for( int k=0; k<6; k++ ) row[k] = pre.row[k];
    LDS_FENCE();   /* the children's write-backs of the previous iteration precede the gathers below */
    if( REC_FLAGS( rec ) & 1 ){
      pr += cpa;
#pragma unroll
      for( int k=0; k<6; k++ ) row[k] += crow[k];
    } else {
      for( int cc=0; cc<REC_NCHILD( rec ); cc++ ){
        const int chp = L.CHP[rec.coff+cc], ch = chp & 0xFF;
        pr += L.PA[6*ch+rr];
        const int ps = ( chp >> 8 ) - 1;
        if( ps >= 0 ){
#pragma unroll
          for( int k=0; k<6; k++ ) row[k] += L.POOL[36*ps+6*rr+k];
        }
      }
    }
    QST(9);
    const double S_r = pre.S_r;
    double u0 = row[0]*pre.S[0], u1 = row[1]*pre.S[1];
    u0 = fma( row[2], pre.S[2], u0 ); u1 = fma( row[3], pre.S[3], u1 );
    u0 = fma( row[4], pre.S[4], u0 ); u1 = fma( row[5], pre.S[5], u1 );
    const double U_r = u0 + u1;
    double dsum = ( on && is1 ) ? S_r*U_r : 0.0, usum = ( on && is1 ) ? S_r*pr : 0.0;
    G8SUM2( dsum, usum );
    const double Dinv = RKFD_RCP( dsum + pre.jm );
    const double u = pre.tau - usum;
    QST(10);
    {
      /* rank-1 downdate Ia = IA - U U'/D: every row needs every U[k] */
      const double tt = is1 ? U_r*Dinv : 0.0;
      const double b0 = G8BCAST( U_r, 0 ), b1 = G8BCAST( U_r, 1 ), b2 = G8BCAST( U_r, 2 );
      const double b3 = G8BCAST( U_r, 3 ), b4 = G8BCAST( U_r, 4 ), b5 = G8BCAST( U_r, 5 );
      row[0] = fma( -tt, b0, row[0] ); row[1] = fma( -tt, b1, row[1] ); row[2] = fma( -tt, b2, row[2] );
      row[3] = fma( -tt, b3, row[3] ); row[4] = fma( -tt, b4, row[4] ); row[5] = fma( -tt, b5, row[5] );
    }
    double pa = pr;
    if( is1 ){
      /* pa = pA + Ia c + U u / D */
      double s0 = row[0]*pre.c[0], s1 = row[1]*pre.c[1];
      s0 = fma( row[2], pre.c[2], s0 ); s1 = fma( row[3], pre.c[3], s1 );
      s0 = fma( row[4], pre.c[4], s0 ); s1 = fma( row[5], pre.c[5], s1 );
      pa = pr + ( s0 + s1 ) + U_r*( u*Dinv );
    } else if( isf ){
      pa = 0;
    }
    QST(11);
    /* write back (needed by later phases and by parents that gather from LDS) */
    if( on ){
      /* Ia goes to LDS only where somebody will read it: a gathering parent (pool slot REC_POOL( rec ))
       * or the Cholesky of a float joint (slot REC_FSLOT( rec )) */
      if( REC_POOL( rec ) >= 0 ){
        /* (a float joint hands nothing to its parent; it owns a pool slot only as a breakable joint that has broken) */
#pragma unroll
        for( int k=0; k<6; k++ ) L.POOL[36*REC_POOL( rec )+6*rr+k] = isf ? 0.0 : row[k];
      }
      if( isf || isb ){
#pragma unroll
        for( int k=0; k<6; k++ ) if( k <= rr ) L.CHOL[21*REC_FSLOT( rec )+RKFD_TRI( rr, k )] = row[k];
      }
      if( isb ) L.XF[12*REC_FSLOT( rec )+rr] = pr;      /* the bias force, for the break test (the frame slot's rotation is not needed today) */
      if( is1 ) L.U[6*i+rr] = U_r;
      L.PA[6*i+rr] = pa;
      if( isf ) L.U[6*i+rr] = pr;   /* float: the U slot keeps the bias pA */
      if( rr == 0 && is1 ){
        L.MS[3*i+0] = Dinv;
        L.MS[3*i+1] = u;
      }
    }
    LDS_FENCE();
    QST(12);
    if( isf && onl && r == 0 ) d_chol6_inplace( &L.CHOL[21*REC_FSLOT( rec )] );
    QST(13);
#undef QST
#pragma unroll
    for( int k=0; k<6; k++ ) crow[k] = row[k];
    cpa = pa;
  }
  SYNC();
}

/* ABA sweep 3 (root to leaf): accelerations and joint accelerations.  Same pipelining; along
 * chains the parent's acceleration stays in registers (schedule flag bit 1).
 * delta = false: the forward part of rkChainUpdateABI, acc = joint accelerations.
 * delta = true : the response to the contact forces found by the MLCP solve, added onto acc -
 *   the same recursion without the velocity-product terms, driven by the innovations the
 *   forces cause (MS slot 1 = du/D of 1-DoF joints, U slot of a float joint = L^-1 of its bias
 *   change).  By linearity of the dynamics in the external forces this equals re-running both
 *   sweeps with the contact wrenches applied (rkChainUpdateCachedABI, reference src/rkfd_mlcp.c:292-296). */
typedef struct { double c_r, U_r, S_r, u, Dinv; } rkfdPre3;
template<bool delta> RKFD_DEV void rkfd_pre3_load(const rkfdLds &L, int i, int rr, rkfdPre3 &p)
{
  p.c_r = delta ? 0.0 : L.C[6*i+rr]; p.U_r = L.U[6*i+rr]; p.S_r = L.S[6*i+rr];
  p.Dinv = L.MS[3*i+0]; p.u = L.MS[3*i+1];
}
template<bool delta> RKFD_DEV void rkfd_phase_sweep3(const rkfdDevModel &m, const rkfdLds &L)
{
  const int lane = LANE();
  const int g = lane >> 3, r = lane & 7;
  const int rr = r < 6 ? r : 0;
  const int T = m.nsched;
  rkfdRec rec1 = rkfd_rec_load( m, 0, g ), rec2 = rkfd_rec_load( m, 1, g );
  double ca = 0;
  for( int t=0; t<T; t++ ){
    const rkfdRec rec = rec1;
    rec1 = rec2;
    rec2 = rkfd_rec_load( m, t+2, g );
    rkfdPre3 pre;
    rkfd_pre3_load<delta>( L, rec.i >= 0 ? rec.i : 0, rr, pre );
    const bool onl = rec.i >= 0;
    const bool on = onl && r < 6;
    const int i = onl ? rec.i : 0;
    const int jt = onl ? RKFD_LI_JT( m.has_brf ? L.LI[i] : rec.li ) : RKFD_JOINT_FIXED;
    const bool is1 = RKFD_JT_IS1( jt );
    const int par = onl ? RKFD_LI_PAR( rec.li ) : -1;
    const int off = RKFD_LI_OFF( rec.li );
    double ap;
    LDS_FENCE();   /* the parents' accelerations written in the previous iteration precede the loads below */
    if( REC_FLAGS( rec ) & 2 ) ap = ca;
    else ap = ( par >= 0 ) ? L.AC[6*par+rr] : 0.0;
    const double y = ap + pre.c_r;
    const double uy = G8SUM( ( on && is1 ) ? pre.U_r*y : 0.0 );
    double a = y;
    if( is1 ){
      const double qdd = delta ? fma( -uy, pre.Dinv, pre.u ) : ( pre.u - uy )*pre.Dinv;
      a = fma( pre.S_r, qdd, y );
      if( on && rr == 0 ){
        L.acc[off] = qdd;     /* delta: the change of the joint acceleration; rkfd_evaluate adds the free one */
      }
    } else if( jt == RKFD_JOINT_FLOAT ){
      if( onl && r == 0 ){
        /* a = IA^-1 ( -pA ); joint acceleration from a - a_parent - c */
        double rhs[6], x[6], d[6], Row[9], p[3];
        if( delta ){
#pragma unroll
          for( int k=0; k<6; k++ ) rhs[k] = L.U[6*i+k];
          double Lr[21];
          d_chol6_load( &L.CHOL[21*REC_FSLOT( rec )], Lr );
          d_chol6_back( Lr, rhs, x );
        } else {
          double yv[6];
#pragma unroll
          for( int k=0; k<6; k++ ) rhs[k] = -L.U[6*i+k];
          double Lr[21];
          d_chol6_load( &L.CHOL[21*REC_FSLOT( rec )], Lr );
          d_chol6_fwd( Lr, rhs, yv );
          d_chol6_back( Lr, yv, x );
        }
#pragma unroll
        for( int k=0; k<6; k++ ){
          L.AC[6*i+k] = x[k];
          d[k] = x[k] - ( par >= 0 ? L.AC[6*par+k] : 0.0 ) - ( delta ? 0.0 : L.C[6*i+k] );
        }
#pragma unroll
        for( int k=0; k<9; k++ ) Row[k] = L.XF[12*REC_FSLOT( rec )+k];
        p[0] = L.XF[12*REC_FSLOT( rec )+9]; p[1] = L.XF[12*REC_FSLOT( rec )+10]; p[2] = L.XF[12*REC_FSLOT( rec )+11];
        /* wdot_j = Row' alpha ; vdot_j = Row' ( a_O - p x alpha ) */
        double t3[3], lin[3], o1[3], o2[3];
        d_cross( p, d, t3 );
        lin[0] = d[3]-t3[0]; lin[1] = d[4]-t3[1]; lin[2] = d[5]-t3[2];
        d_tmulv( Row, lin, o1 ); d_tmulv( Row, d, o2 );
        L.acc[off] = o1[0]; L.acc[off+1] = o1[1]; L.acc[off+2] = o1[2];
        L.acc[off+3] = o2[0]; L.acc[off+4] = o2[1]; L.acc[off+5] = o2[2];
      }
    }
    if( m.has_brf && on && L.BRK[i] == RKFD_BRF_ATTACHED ) L.acc[off+rr] = 0.0;      /* rigidly attached: its six joint accelerations are zero */
    LDS_FENCE();
    if( jt == RKFD_JOINT_FLOAT ) a = L.AC[6*i+rr];
    if( on && jt != RKFD_JOINT_FLOAT ) L.AC[6*i+rr] = a;
    ca = a;
  }
  SYNC();
}

#endif /* RKFD_DEV_SWEEPS_H */
