/* rkfd_dev_volume.h - part of the device code of the batched rkFDUpdate step (see rkfd_device.h):
 * the Volume plugin's rigid branch (reference src/rkfd_volume.c; restated on the CPU in oracle/rkfd_oracle_volume.h, whose
 * header lists what is taken from un-vendored RoKi / ZM and how).  Only the kernel variant sv == 2 carries this code.
 *
 *   rkfd_phase_volcol  (before the sweeps, while the link frames and velocities are alive)
 *     per rigid pair of convex shapes: vertex test, intersection volume (lane = face of either shape, clipped in place in
 *     LDS against the planes of the other: Sutherland-Hodgman), centre / normal / frame, the area integrals of
 *     _rkFDSolverConstraint (:397-491) per fan triangle, the contact-plane conditions (merged and sorted by one lane:
 *     a handful), the 6-D relative velocity and the sliding velocities at the polygon's corners.
 *   rkfd_phase_volume  (after sweeps 2 and 3, in place of the MLCP phase)
 *     b and A by the same innovation probes as the MLCP phase, six columns per pair (unit world force / torque at the
 *     centre); the QP of _rkFDSolverQPCreate (:496-528); the active-set method of src/rkfd_opt_qp.c with dense constraint
 *     rows (KKT solves through the Cholesky factor of Q and a Jacobi pseudo-inverse of the small Schur complement, which
 *     gives the minimum-norm multipliers of zLESolveMP); centre-of-normal-force and friction fix-ups (:580-631, :869-916)
 *     with a wave-cooperative tableau simplex (lane = column); the inputs of the delta sweep.
 * Included by rkfd_device.h only, in this order; compiles for gfx950 and under the lane emulator. */
#ifndef RKFD_DEV_VOLUME_H
#define RKFD_DEV_VOLUME_H

#define RKFD_VOL_NRED 16
/* sum NV values of the first F lanes, in lane order: scr holds [NV][F] + [NV] doubles; the sums are left in scr[NV*F ..] */
template<int NV> RKFD_DEV void rkfd_vol_reduce(double *scr, const double *val, int F)
{
  const int lane = LANE();
  if( lane < F ){
#pragma unroll
    for( int k=0; k<NV; k++ ) scr[k*F+lane] = val[k];
  }
  SYNC();
  if( lane < NV ){
    double s = 0;
    for( int l=0; l<F; l++ ) s += scr[lane*F+l];
    scr[NV*F+lane] = s;
  }
  SYNC();
}

/* world frame of device link i from the frame arrays of the kinematics phase */
RKFD_DEV void d_vol_frame(const rkfdLds &L, int i, double *R, double *p)
{
#pragma unroll
  for( int k=0; k<6; k++ ) R[k] = L.XA[6*i+k];
#pragma unroll
  for( int k=0; k<3; k++ ){ R[6+k] = L.XB[6*i+k]; p[k] = L.XB[6*i+3+k]; }
}

/* one polygon (n vertices at P, capacity cap) against the half space pl.x - d <= 0, in place (Sutherland-Hodgman: the
 * vertex read next is always ahead of the one written).  ovf is set when a vertex was due beyond the capacity (vol_pv = the
 * longest face loop + the faces of the larger shape bounds a convex polygon clipped by that many planes, so this should be
 * unreachable - but a truncated polygon would silently change centre, normal and Q, so it is reported like every other
 * capacity: status 2). */
RKFD_DEV int d_vol_clip(double *P, int n, int cap, const double *pl, double d, bool &ovf)
{
  if( n < 1 ) return 0;
  /* most planes of the other shape do not touch a given face: look first, rewrite the polygon only when the plane cuts it */
  {
    bool out = false, in = false;
    for( int i=0; i<n; i++ ){
      const double si = pl[0]*P[3*i] + pl[1]*P[3*i+1] + pl[2]*P[3*i+2] - d;
      out = out || si > 0; in = in || si <= 0;
    }
    if( !out ) return n;
    if( !in ) return 0;
  }
  const double f0 = P[0], f1 = P[1], f2 = P[2], sf = pl[0]*f0 + pl[1]*f1 + pl[2]*f2 - d;
  double c0 = f0, c1 = f1, c2 = f2, sc = sf;
  int k = 0;
  for( int i=0; i<n; i++ ){
    double n0 = f0, n1 = f1, n2 = f2, sn = sf;
    if( i+1 < n ){ n0 = P[3*i+3]; n1 = P[3*i+4]; n2 = P[3*i+5]; sn = pl[0]*n0 + pl[1]*n1 + pl[2]*n2 - d; }
    if( sc <= 0 ){ if( k < cap ){ P[3*k] = c0; P[3*k+1] = c1; P[3*k+2] = c2; k++; } else ovf = true; }
    if( ( sc < 0 && sn > 0 ) || ( sc > 0 && sn < 0 ) ){
      if( k < cap ){
        const double t = sc*RKFD_RCP( sc - sn );
        P[3*k] = c0 + t*( n0-c0 ); P[3*k+1] = c1 + t*( n1-c1 ); P[3*k+2] = c2 + t*( n2-c2 );
        k++;
      } else ovf = true;
    }
    c0 = n0; c1 = n1; c2 = n2; sc = sn;
  }
  return k;
}

/* ---- the per-triangle pieces of _rkFDSolverConstraint (reference src/rkfd_volume.c:232-348) ---- */
RKFD_DEV double d_vol_area(const double *p)   /* p: 3 x 3 */
{
  const double e1[3] = { p[3]-p[0], p[4]-p[1], p[5]-p[2] }, e2[3] = { p[6]-p[0], p[7]-p[1], p[8]-p[2] };
  double x[3];
  d_cross( e1, e2, x );
  return 0.5*sqrt( d_dot( x, x ) );
}
RKFD_DEV void d_vol_mid(const double *p, double *pm)
{
#pragma unroll
  for( int k=0; k<3; k++ ){ pm[k] = 0.5*( p[k]+p[3+k] ); pm[3+k] = 0.5*( p[3+k]+p[6+k] ); pm[6+k] = 0.5*( p[6+k]+p[k] ); }
}
/* _rkFDSolverConstraintDepth (:296-310); scales pm in place as the reference does */
RKFD_DEV void d_vol_depth(double *pm, const double *h, double K, double s, const double *norm, double *cc)
{
  const double k = K*s/6.0;
  const double hm[3] = { k*( h[0]+h[1] ), k*( h[1]+h[2] ), k*( h[0]+h[2] ) }, hc = k*( h[0]+h[1]+h[2] )*2;
  cc[0] = -hc*norm[0]; cc[1] = -hc*norm[1]; cc[2] = -hc*norm[2]; cc[3] = cc[4] = cc[5] = 0;
#pragma unroll
  for( int i=0; i<3; i++ ){
    double t[3];
    pm[3*i] *= hm[i]; pm[3*i+1] *= hm[i]; pm[3*i+2] *= hm[i];
    d_cross( norm, &pm[3*i], t );
    cc[3] += t[0]; cc[4] += t[1]; cc[5] += t[2];
  }
}
RKFD_DEV void d_vol_inner(const double *p1, const double *p2, double h1, double h2, double *pp)
{
  if( fabs( h1 ) < RKFD_DEV_TOL ){ pp[0] = p1[0]; pp[1] = p1[1]; pp[2] = p1[2]; return; }
  if( fabs( h2 ) < RKFD_DEV_TOL ){ pp[0] = p2[0]; pp[1] = p2[1]; pp[2] = p2[2]; return; }
  if( fabs( h2 - h1 ) < RKFD_DEV_TOL ){ pp[0] = 0.5*( p1[0]+p2[0] ); pp[1] = 0.5*( p1[1]+p2[1] ); pp[2] = 0.5*( p1[2]+p2[2] ); return; }
  const double a = h2*RKFD_RCP( h2 - h1 ), b = h1*RKFD_RCP( h1 - h2 );
  pp[0] = a*p1[0]; pp[1] = a*p1[1]; pp[2] = a*p1[2];
  pp[0] += b*p2[0]; pp[1] += b*p2[1]; pp[2] += b*p2[2];
}
/* the face's own contact-plane condition (_rkFDSolverSetContactPlane, :350-374, restricted to one face: all its
 * conditions have the same direction and lie on one line, so only the extreme point survives) */
typedef struct { int has; double v[3], n[3]; } rkfdVolCP;
RKFD_DEV void d_vol_set_plane(rkfdVolCP &cp, const double *p, const double *fnorm, const double *norm)
{
  const double dn = d_dot( fnorm, norm );
  const double t[3] = { fnorm[0]-dn*norm[0], fnorm[1]-dn*norm[1], fnorm[2]-dn*norm[2] };
  if( fabs( t[0] ) < RKFD_DEV_TOL && fabs( t[1] ) < RKFD_DEV_TOL && fabs( t[2] ) < RKFD_DEV_TOL ) return;
  const double l = sqrt( d_dot( t, t ) );
  const double nn[3] = { t[0]*( -1.0/l ), t[1]*( -1.0/l ), t[2]*( -1.0/l ) };
  if( !cp.has ){
    cp.has = 1;
    cp.v[0] = p[0]; cp.v[1] = p[1]; cp.v[2] = p[2]; cp.n[0] = nn[0]; cp.n[1] = nn[1]; cp.n[2] = nn[2];
    return;
  }
  const double d[3] = { cp.v[0]-p[0], cp.v[1]-p[1], cp.v[2]-p[2] };
  if( !( fabs( d_dot( nn, d ) ) < 1e-8 ) ) return;
  double x[3];
  d_cross( nn, d, x );
  if( d_dot( norm, x ) > 0.0 ){ cp.v[0] = p[0]; cp.v[1] = p[1]; cp.v[2] = p[2]; }
}

/* one triangle of the intersection volume (:408-488): acc[0] += area, acc[1..3] += area-weighted centroid, acc[4..9] +=
 * (s/3) sum [pm x][pm x] (xx,xy,xz,yy,yz,zz), acc[10..15] += c */
RKFD_DEV void d_vol_triangle(const double *tri, const double *fnorm, const double *center, const double *norm, double K, double *acc, rkfdVolCP &cp)
{
  double h[3], pf[9], p[9], pm[9], cc[6];
#pragma unroll
  for( int j=0; j<3; j++ ){
    pf[3*j] = tri[3*j]-center[0]; pf[3*j+1] = tri[3*j+1]-center[1]; pf[3*j+2] = tri[3*j+2]-center[2];
    h[j] = d_dot( norm, &pf[3*j] );
    p[3*j] = pf[3*j]-h[j]*norm[0]; p[3*j+1] = pf[3*j+1]-h[j]*norm[1]; p[3*j+2] = pf[3*j+2]-h[j]*norm[2];
  }
  d_vol_mid( p, pm );
  double s = d_vol_area( p );
  {
    /* _rkFDSolverConstraintAddQ (:279-294) */
    const double k = s/3.0;
    acc[0] += s;
    acc[1] += k*( p[0]+p[3]+p[6] ); acc[2] += k*( p[1]+p[4]+p[7] ); acc[3] += k*( p[2]+p[5]+p[8] );
    double mm[6] = {0,0,0,0,0,0};
#pragma unroll
    for( int i=0; i<3; i++ ){
      const double v0 = pm[3*i], v1 = pm[3*i+1], v2 = pm[3*i+2];
      mm[0] += -v2*v2 - v1*v1; mm[1] += v1*v0; mm[2] += v2*v0;
      mm[3] += -v2*v2 - v0*v0; mm[4] += v2*v1; mm[5] += -v1*v1 - v0*v0;
    }
#pragma unroll
    for( int i=0; i<6; i++ ) acc[4+i] += k*mm[i];
  }
  d_vol_depth( pm, h, K, s, norm, cc );
  int st = 0, s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
  for( int j=0; j<3; j++ ){
    if( h[j] > RKFD_DEV_TOL ){ st += 1 << ( j*2 ); s1 = j; }
    else if( h[j] < -RKFD_DEV_TOL ){ st += 1 << ( j*2+1 ); s2 = j; }
    else s0 = j;
  }
  /* The reference's switch (:422-475) indexes the vertices by their roles; with run-time indices the arrays would live in the
   * private segment (scratch memory).  Here the triangle is ROTATED (a cyclic shift keeps its orientation) so that the vertex
   * with the special role comes first; the roles of the other two are then fixed positions:
   *   one on the plane, one above, one below  (0x24 0x12 0x09 | 0x06 0x21 0x18): on-plane vertex first -> ( on, above, below ) | ( on, below, above )
   *   two above, one below (0x16 0x19 0x25): the one below first;   two below, one above (0x1a 0x26 0x29): the one above first */
  const int nab = ( st & 0x15 ) ? __builtin_popcount( st & 0x15 ) : 0, nbl = __builtin_popcount( st & 0x2a );
  if( nab + nbl == 0 ) return;                                         /* all on the plane */
  if( nbl == 0 || nab == 0 ){
    /* one side only: the faces' conditions where a vertex lies on the plane; c +- cc */
    if( nab + nbl < 3 ){
      const double q[3] = { s0 == 0 ? pf[0] : ( s0 == 1 ? pf[3] : pf[6] ), s0 == 0 ? pf[1] : ( s0 == 1 ? pf[4] : pf[7] ), s0 == 0 ? pf[2] : ( s0 == 1 ? pf[5] : pf[8] ) };
      d_vol_set_plane( cp, q, fnorm, norm );
    }
    const double sg = nbl == 0 ? 1.0 : -1.0;
#pragma unroll
    for( int k=0; k<6; k++ ) acc[10+k] += sg*cc[k];
    return;
  }
  /* straddling: c +- cc, then -+ 2 x the part cut off */
  const bool ab = nab == 1 && nbl == 1;                 /* one on, one above, one below */
  const bool dd = !ab && nbl == 2;                      /* two below, one above */
  const int rot = ab ? s0 : ( dd ? s1 : s2 );
  {
    const double sg = dd ? -1.0 : 1.0;
#pragma unroll
    for( int k=0; k<6; k++ ) acc[10+k] += sg*cc[k];
  }
  double qf[9], qp[9], qh[3];
#pragma unroll
  for( int j=0; j<3; j++ ){
    /* position j of the rotated triangle = vertex ( rot + j ) % 3 */
#pragma unroll
    for( int k=0; k<3; k++ ){
      const double f0 = pf[3*j+k], f1 = pf[3*( ( j+1 )%3 )+k], f2 = pf[3*( ( j+2 )%3 )+k];
      const double p0 = p[3*j+k], p1 = p[3*( ( j+1 )%3 )+k], p2 = p[3*( ( j+2 )%3 )+k];
      qf[3*j+k] = rot == 0 ? f0 : ( rot == 1 ? f1 : f2 );
      qp[3*j+k] = rot == 0 ? p0 : ( rot == 1 ? p1 : p2 );
    }
    const double h0 = h[j], h1 = h[( j+1 )%3], h2 = h[( j+2 )%3];
    qh[j] = rot == 0 ? h0 : ( rot == 1 ? h1 : h2 );
  }
  double pp0[3];
  if( ab ){
    /* ( on, above, below ): the cut replaces the ABOVE vertex (position 1) and the part kept is the one below;
     * ( on, below, above ): the cut replaces the BELOW vertex (position 1 again) and the part kept is the one above - as written
     * in the reference (:435-452); both take the cut point between the vertex above and the vertex below, in that order */
    const bool bt = qh[1] < 0;
    double u[3], v[3];
#pragma unroll
    for( int k=0; k<3; k++ ){ u[k] = bt ? qf[6+k] : qf[3+k]; v[k] = bt ? qf[3+k] : qf[6+k]; }
    d_vol_inner( u, v, bt ? qh[2] : qh[1], bt ? qh[1] : qh[2], &qp[3] );
    qh[1] = 0.0;
#pragma unroll
    for( int k=0; k<3; k++ ) pp0[k] = bt ? qp[3+k] : qp[k];
  } else {
    /* the lone vertex first: both others are replaced by the cut points of its edges */
    d_vol_inner( qf, &qf[3], qh[0], qh[1], &qp[3] );
    d_vol_inner( qf, &qf[6], qh[0], qh[2], &qp[6] );
    qh[1] = qh[2] = 0.0;
#pragma unroll
    for( int k=0; k<3; k++ ) pp0[k] = dd ? qp[6+k] : qp[3+k];
  }
  s = d_vol_area( qp );
  d_vol_mid( qp, pm );
  d_vol_depth( pm, qh, K, s, norm, cc );
  {
    const double sgn = dd ? 2.0 : -2.0;
#pragma unroll
    for( int k=0; k<6; k++ ) acc[10+k] += sgn*cc[k];
  }
  d_vol_set_plane( cp, pp0, fnorm, norm );
}

/* the packed pair record of the device model */
#define RKFD_VP_LA(r)  ( (r)[0] )
#define RKFD_VP_LB(r)  ( (r)[1] )
#define RKFD_VP_CI(r)  ( (r)[2] )
#define RKFD_VP_A0(r)  ( (r)[3] )
#define RKFD_VP_NA(r)  ( (r)[4] )
#define RKFD_VP_B0(r)  ( (r)[5] )
#define RKFD_VP_NB(r)  ( (r)[6] )
/* per colliding pair, in L.VD (stride RKFD_VD): centre 3, axes 9, accumulated area integrals 16, relative velocity
 * (lin 3, ang 3), velocity-product part of the relative point acceleration 3, tangential velocity at the centre 2,
 * wrench 6 */
#define RKFD_VD      48
#define RKFD_VD_C    0
#define RKFD_VD_AX   3
#define RKFD_VD_ACC  12
#define RKFD_VD_VEL  28
#define RKFD_VD_CA   34
#define RKFD_VD_TC   37
#define RKFD_VD_W    39
/* per contact-plane condition, in L.VPL (stride 8): point 3, inward normal 3, tangential velocity there 2 */

/* ------------------------------------------------------------------------ */
template<bool prof> RKFD_DEV void rkfd_phase_volcol(const rkfdDevModel &m, const rkfdLds &L, unsigned long long *pc)
{
  unsigned long long q0 = prof ? RKFD_CLOCK() : 0ull, q1;
#define VCT(k) do{ if( prof ){ q1 = RKFD_CLOCK(); pc[k] += q1 - q0; q0 = q1; } }while(0)
  const int lane = LANE();
  const int PV = m.vol_pv, NCP = m.vol_ncp;
  const int F = m.vol_nf;
  double *P = &L.VPOLY[( lane < F ? lane : 0 )*PV*3];
  int nvp = 0;
  for( int pr=0; pr<m.vol_npair; pr++ ){
    const int *rec = &RELOAD( m.vol_pair )[8*pr];
    const int la = RKFD_VP_LA( rec ), lb = RKFD_VP_LB( rec ), ci = RKFD_VP_CI( rec );
    const int a0 = RKFD_VP_A0( rec ), na = RKFD_VP_NA( rec ), b0 = RKFD_VP_B0( rec ), nb = RKFD_VP_NB( rec );
    double RA[9], pA[3], RB[9], pB[3];
    d_vol_frame( L, la, RA, pA ); d_vol_frame( L, lb, RB, pB );
    /* "Vert": some vertex of one shape inside the other */
    int hit = 0;
    for( int side=0; side<2; side++ ){
      const int f0 = side == 0 ? a0 : b0, nf = side == 0 ? na : nb, g0 = side == 0 ? b0 : a0, ng = side == 0 ? nb : na;
      const double *Rs = side == 0 ? RA : RB, *ps = side == 0 ? pA : pB, *Ro = side == 0 ? RB : RA, *po = side == 0 ? pB : pA;
      const int v0 = RELOAD( m.vol_loop )[2*f0], v1 = RELOAD( m.vol_loop )[2*( f0+nf-1 )] + RELOAD( m.vol_loop )[2*( f0+nf-1 )+1];
      for( int vb=v0; vb<v1; vb+=RKFD_WAVE ){
        const int v = vb + lane;
        int in = 0;
        if( v < v1 ){
          const double *vl = &RELOAD( m.vol_lvert )[3*v];
          double x[3], r[3], y[3], smax = -HUGE_VAL;
          d_mulv( Rs, vl, x );
          r[0] = x[0]+ps[0]-po[0]; r[1] = x[1]+ps[1]-po[1]; r[2] = x[2]+ps[2]-po[2];
          d_tmulv( Ro, r, y );
          for( int g=g0; g<g0+ng; g++ ){
            const double *pl = &RELOAD( m.vol_lplane )[4*g];
            const double s = pl[0]*y[0] + pl[1]*y[1] + pl[2]*y[2] - pl[3];
            if( s > smax ) smax = s;
          }
          in = smax < RKFD_DEV_TOL;
        }
        if( BALLOT( in ) != 0ull ) hit = 1;
      }
    }
    VCT(28);
    if( !hit ) continue;
    if( rec[7] & 4 ){ if( lane == 0 ) L.cnt[CNT_GRD] = 1; continue; }      /* a pair that cannot be clipped is in contact: status 4 */
    if( nvp >= m.vol_np ){ if( lane == 0 ) L.cnt[CNT_OVF] = 1; continue; }
    /* faces of A inside B, faces of B inside A: lane = face */
    const bool onf = lane < na+nb;
    const bool isA = lane < na;
    const int f = onf ? ( isA ? a0+lane : b0+lane-na ) : 0;
    const double *Rs = isA ? RA : RB, *ps = isA ? pA : pB;
    int n = 0;
    double nw[3] = {0,0,0};
    /* every face's plane in world coordinates, staged in LDS: the clipping loops below read the other shape's planes from there */
    double *stg = L.VRED;
    if( onf ){
      const double *pl = &RELOAD( m.vol_lplane )[4*f];
      d_mulv( Rs, pl, nw );
      stg[4*lane] = nw[0]; stg[4*lane+1] = nw[1]; stg[4*lane+2] = nw[2]; stg[4*lane+3] = pl[3] + d_dot( nw, ps );
    }
    SYNC();
    if( onf ){
      const int v0 = RELOAD( m.vol_loop )[2*f];
      n = RELOAD( m.vol_loop )[2*f+1];
      for( int i=0; i<n; i++ ){
        double x[3];
        d_mulv( Rs, &RELOAD( m.vol_lvert )[3*( v0+i )], x );
        P[3*i] = x[0]+ps[0]; P[3*i+1] = x[1]+ps[1]; P[3*i+2] = x[2]+ps[2];
      }
      const int g0 = isA ? na : 0, ng = isA ? nb : na;      /* the other shape's faces are lanes g0 .. g0+ng-1 */
      bool ovf = false;
      for( int g=g0; g<g0+ng && n>=3; g++ ) n = d_vol_clip( P, n, PV, &stg[4*g], stg[4*g+3], ovf );
      if( ovf ) L.cnt[CNT_OVF] = 1;      /* (lanes that write all write the same value) */
      if( n < 3 ) n = 0;
    }
    /* reference point of the signed tetrahedra: the first vertex of the first face that survived (a point ON the volume: with
     * a far one the centroid of a thin slab loses five digits to cancellation, and the depths h are measured from it) */
    double ref[3] = {0,0,0};
    {
      SYNC();
      const unsigned long long mk = BALLOT( n >= 3 );
      if( mk ){ const double *P0 = &L.VPOLY[__builtin_ctzll( mk )*PV*3]; ref[0] = P0[0]; ref[1] = P0[1]; ref[2] = P0[2]; }
    }
    double val[RKFD_VOL_NRED];
#pragma unroll
    for( int k=0; k<RKFD_VOL_NRED; k++ ) val[k] = 0;
    for( int i=1; i+1<n; i++ ){
      const double e1[3] = { P[3*i]-P[0], P[3*i+1]-P[1], P[3*i+2]-P[2] }, e2[3] = { P[3*i+3]-P[0], P[3*i+4]-P[1], P[3*i+5]-P[2] };
      double x[3];
      d_cross( e1, e2, x );
      if( sqrt( d_dot( x, x ) ) < 1e-24 ) continue;
      val[0] += 1.0;
      if( !isA ){ val[1] += 0.5*x[0]; val[2] += 0.5*x[1]; val[3] += 0.5*x[2]; }
      const double a[3] = { P[0]-ref[0], P[1]-ref[1], P[2]-ref[2] }, b[3] = { P[3*i]-ref[0], P[3*i+1]-ref[1], P[3*i+2]-ref[2] };
      const double c[3] = { P[3*i+3]-ref[0], P[3*i+4]-ref[1], P[3*i+5]-ref[2] };
      double y[3];
      d_cross( b, c, y );
      const double w = d_dot( a, y );
      val[4] += w;
      val[5] += w*( a[0]+b[0]+c[0] ); val[6] += w*( a[1]+b[1]+c[1] ); val[7] += w*( a[2]+b[2]+c[2] );
    }
    double *scr = L.VRED;
    rkfd_vol_reduce<8>( scr, val, F );
    const double *sum = &scr[8*F];
    const double ntri = sum[0], asB[3] = { sum[1], sum[2], sum[3] }, v6 = sum[4];
    const double lasB = sqrt( d_dot( asB, asB ) );
    const bool ok = ntri >= 4.0 && v6 > 1e-30 && lasB > 1e-30;
    double center[3] = {0,0,0}, ax[9];
#pragma unroll
    for( int k=0; k<9; k++ ) ax[k] = 0;
    if( ok ){
      center[0] = ref[0] + sum[5]/( 4.0*v6 ); center[1] = ref[1] + sum[6]/( 4.0*v6 ); center[2] = ref[2] + sum[7]/( 4.0*v6 );
      ax[0] = asB[0]*( 1.0/lasB ); ax[1] = asB[1]*( 1.0/lasB ); ax[2] = asB[2]*( 1.0/lasB );
      d_ortho_space( ax, ax+3, ax+6 );
    }
    SYNC();
    VCT(29);
    if( !ok ) continue;
    /* area integrals and the faces' own contact-plane conditions */
    rkfdVolCP cp; cp.has = 0;
    cp.v[0] = cp.v[1] = cp.v[2] = 0; cp.n[0] = cp.n[1] = cp.n[2] = 0;
#pragma unroll
    for( int k=0; k<RKFD_VOL_NRED; k++ ) val[k] = 0;
    {
      const double K = m.ci_k[ci];
      for( int i=1; i+1<n; i++ ){
        const double tri[9] = { P[0], P[1], P[2], P[3*i], P[3*i+1], P[3*i+2], P[3*i+3], P[3*i+4], P[3*i+5] };
        const double e1[3] = { tri[3]-tri[0], tri[4]-tri[1], tri[5]-tri[2] }, e2[3] = { tri[6]-tri[0], tri[7]-tri[1], tri[8]-tri[2] };
        double x[3];
        d_cross( e1, e2, x );
        if( sqrt( d_dot( x, x ) ) < 1e-24 ) continue;
        d_vol_triangle( tri, nw, center, ax, K, val, cp );
      }
    }
    SYNC();
    rkfd_vol_reduce<16>( scr, val, F );
    {
      double *vd = &L.VD[RKFD_VD*nvp];
      if( lane < 3 ) vd[RKFD_VD_C+lane] = center[lane];
      if( lane < 9 ) vd[RKFD_VD_AX+lane] = ax[lane];
      if( lane < 16 ) vd[RKFD_VD_ACC+lane] = scr[16*F+lane];
      if( lane < 6 ) vd[RKFD_VD_W+lane] = 0.0;
    }
    SYNC();
    VCT(30);
    /* gather the conditions in face order, merge identical ones, sort by angle (:350-395, :490): one lane, a handful of entries */
    {
      double *cv = scr;          /* [F][8]: has, v, n, angle per face (the angle of __rk_fd_plane_cmp, :376-395, by the face's own lane); then [F] angles of the list */
      if( lane < F ){
        cv[8*lane] = cp.has ? 1.0 : 0.0;
#pragma unroll
        for( int k=0; k<3; k++ ){ cv[8*lane+1+k] = cp.v[k]; cv[8*lane+4+k] = cp.n[k]; }
        double tmp[3];
        d_cross( ax+3, cp.n, tmp );
        const double y = sqrt( d_dot( tmp, tmp ) ), a = cp.has ? d_atan2_ypos( y, d_dot( ax+3, cp.n ) ) : 0.0;
        cv[8*lane+7] = d_dot( tmp, ax ) > 0 ? -a : a;
      }
      SYNC();
      if( lane == 0 ){
        double *pl = &L.VPL[8*NCP*nvp];
        double *th = scr + 8*F;
        int ncp = 0, ovf = 0;
        for( int fi=0; fi<na+nb; fi++ ){
          if( cv[8*fi] == 0.0 ) continue;
          const double *p = &cv[8*fi+1], *nn = &cv[8*fi+4];
          int merged = 0;
          for( int k=0; k<ncp && !merged; k++ ){
            double *c2 = &pl[8*k];
            const double dn[3] = { nn[0]-c2[3], nn[1]-c2[4], nn[2]-c2[5] };
            if( !( fabs( dn[0] ) < 1e-8 && fabs( dn[1] ) < 1e-8 && fabs( dn[2] ) < 1e-8 ) ) continue;
            const double d[3] = { c2[0]-p[0], c2[1]-p[1], c2[2]-p[2] };
            if( !( fabs( d_dot( nn, d ) ) < 1e-8 ) ) continue;
            double x[3];
            d_cross( nn, d, x );
            if( d_dot( ax, x ) > 0.0 ){ c2[0] = p[0]; c2[1] = p[1]; c2[2] = p[2]; }
            merged = 1;
          }
          if( merged ) continue;
          if( ncp == NCP ){ ovf = 1; continue; }
          double *c2 = &pl[8*ncp];
          c2[0] = p[0]; c2[1] = p[1]; c2[2] = p[2]; c2[3] = nn[0]; c2[4] = nn[1]; c2[5] = nn[2];
          th[ncp] = cv[8*fi+7];
          ncp++;
        }
        for( int i=1; i<ncp; i++ ){
          double x[6], t = th[i];
          int j;
#pragma unroll
          for( int k=0; k<6; k++ ) x[k] = pl[8*i+k];
          for( j=i-1; j>=0 && !( fabs( th[j] - t ) < RKFD_DEV_TOL ) && th[j] > t; j-- ){
#pragma unroll
            for( int k=0; k<6; k++ ) pl[8*( j+1 )+k] = pl[8*j+k];
            th[j+1] = th[j];
          }
#pragma unroll
          for( int k=0; k<6; k++ ) pl[8*( j+1 )+k] = x[k];
          th[j+1] = t;
        }
        L.VI[2*nvp] = ncp; L.VI[2*nvp+1] = pr;
        if( ovf ) L.cnt[CNT_OVF] = 1;
      }
      SYNC();
    }
    /* velocities (the link velocities die with sweep 2): 6-D relative velocity at the centre (rkFDChainPointRelativeVel6D),
     * the velocity-product part of the relative point acceleration, tangential velocities at the centre and at the corners
     * of the contact polygon (rkFDChainPointRelativeVel less its normal part; :715-731, :759-776) */
    {
      const int ncp = L.VI[2*nvp];
      double *vd = &L.VD[RKFD_VD*nvp];
      if( lane <= ncp ){
        double x[3] = { center[0], center[1], center[2] };
        if( lane > 0 ){ const double *c2 = &L.VPL[8*( NCP*nvp + lane-1 )]; x[0] += c2[0]; x[1] += c2[1]; x[2] += c2[2]; }
        double va[3], vb[3];
        d_point_vel( &L.V[6*la], x, va ); d_point_vel( &L.V[6*lb], x, vb );
        double v[3] = { va[0]-vb[0], va[1]-vb[1], va[2]-vb[2] };
        double vs[3] = { v[0], v[1], v[2] };          /* with the cells' slide velocities (the friction fix-ups; not the 6-D velocity) */
        if( rec[7] & 3 ){
#pragma unroll
          for( int sd=0; sd<2; sd++ ){
            if( !( ( rec[7] >> sd ) & 1 ) ) continue;
            const double *sp = &RELOAD( m.vol_slide )[16*pr + 8*sd];
            const double *Rk = sd == 0 ? RA : RB, *pk = sd == 0 ? pA : pB;
            const double axl[3] = { sp[1], sp[2], sp[3] }, orl[3] = { sp[4], sp[5], sp[6] };
            double axw[3], orw[3], sv[3];
            d_mulv( Rk, axl, axw ); d_mulv( Rk, orl, orw );
            const double t[3] = { x[0]-pk[0]-orw[0], x[1]-pk[1]-orw[1], x[2]-pk[2]-orw[2] };
            d_cross( axw, t, sv );
            const double sn = d_dot( sv, ax );
            sv[0] -= sn*ax[0]; sv[1] -= sn*ax[1]; sv[2] -= sn*ax[2];
            const double nr = sqrt( d_dot( sv, sv ) );
            if( fabs( nr ) < RKFD_DEV_TOL ) continue;
            const double gq = ( sd == 0 ? 1.0 : -1.0 )*sp[0]/nr;
            vs[0] += gq*sv[0]; vs[1] += gq*sv[1]; vs[2] += gq*sv[2];
          }
        }
        if( lane == 0 ){
          double ca[3], cb[3];
          d_cross( &L.V[6*la], va, ca ); d_cross( &L.V[6*lb], vb, cb );
#pragma unroll
          for( int k=0; k<3; k++ ){
            vd[RKFD_VD_VEL+k] = v[k]; vd[RKFD_VD_VEL+3+k] = L.V[6*la+k] - L.V[6*lb+k];
            vd[RKFD_VD_CA+k] = ca[k] - cb[k];
          }
        }
        const double vn = d_dot( ax, vs );
        vs[0] -= vn*ax[0]; vs[1] -= vn*ax[1]; vs[2] -= vn*ax[2];
        const double t1 = d_dot( vs, ax+3 ), t2 = d_dot( vs, ax+6 );
        if( lane == 0 ){ vd[RKFD_VD_TC] = t1; vd[RKFD_VD_TC+1] = t2; }
        else { double *c2 = &L.VPL[8*( NCP*nvp + lane-1 )]; c2[6] = t1; c2[7] = t2; }
      }
    }
    SYNC();
    VCT(31);
    nvp++;
  }
  if( lane == 0 ) L.cnt[CNT_NVP] = nvp;
  SYNC();
#undef VCT
}

/* ------------------------------------------------------------------------ */
/* Moore-Penrose solve of a small symmetric positive semi-definite system in LDS, wave-cooperative cyclic Jacobi
 * (the same method and rank tolerance as sym_pinv_solve of the oracle): S (r x r, stride ld, destroyed), EV (r x r, stride ld)
 * scratch; lane i passes rhs_i and receives x_i. */
RKFD_DEV double rkfd_vol_pinv(double *S, double *EV, double *vec, int ld, int r, double rhs_i)
{
  const int lane = LANE();
  if( r == 0 ) return 0.0;
  if( r == 1 ){
    /* one active row: x = rhs / s unless s is (numerically) zero */
    const double s00 = S[0], r0 = BCAST( rhs_i, 0 );
    return ( lane == 0 && fabs( s00 ) > 0.0 ) ? r0/s00 : 0.0;
  }
  if( r == 2 ){
    /* two active rows (the centre of normal force at a corner of the polygon): ONE Jacobi rotation diagonalises a 2 x 2 matrix -
     * the same rotation, eigenvalues and rank test as the general loop below, in scalar arithmetic without barriers */
    const double a = S[0], bq = S[1], d = S[ld+1], r0 = BCAST( rhs_i, 0 ), r1 = BCAST( rhs_i, 1 );
    double c = 1.0, sn = 0.0, e0 = a, e1 = d;
    if( bq != 0.0 ){
      const double theta = ( d - a )/( 2.0*bq );
      const double t = ( theta >= 0 ? 1.0 : -1.0 )/( fabs( theta ) + sqrt( theta*theta + 1.0 ) );
      c = 1.0/sqrt( t*t + 1.0 ); sn = t*c;
      /* the diagonal after the column and the row rotation of the loop below */
      const double k00 = c*a - sn*bq, k01 = sn*a + c*bq, k10 = c*bq - sn*d, k11 = sn*bq + c*d;
      e0 = c*k00 - sn*k10; e1 = sn*k01 + c*k11;
    }
    /* eigenvectors: columns ( c, -sn ) and ( sn, c ) */
    const double wmax = fabs( e0 ) > fabs( e1 ) ? fabs( e0 ) : fabs( e1 );
    const double y0 = fabs( e0 ) > 1e-12*wmax ? ( c*r0 - sn*r1 )/e0 : 0.0, y1 = fabs( e1 ) > 1e-12*wmax ? ( sn*r0 + c*r1 )/e1 : 0.0;
    return lane == 0 ? c*y0 + sn*y1 : ( lane == 1 ? -sn*y0 + c*y1 : 0.0 );
  }
  if( lane < r ) for( int j=0; j<r; j++ ) EV[lane*ld+j] = lane == j ? 1.0 : 0.0;
  SYNC();
  for( int sweep=0; sweep<60; sweep++ ){
    double off = 0, dg = 0;
    if( lane < r ){
      dg = S[lane*ld+lane]*S[lane*ld+lane];
      for( int j=lane+1; j<r; j++ ) off += S[lane*ld+j]*S[lane*ld+j];
    }
    const double soff = WSUM( off ), sdg = WSUM( dg );
    if( soff <= 1e-30*( sdg + soff ) || soff == 0 ) break;
    for( int i=0; i<r-1; i++ )
      for( int j=i+1; j<r; j++ ){
        const double apq = S[i*ld+j];
        if( apq == 0.0 ) continue;            /* (uniform: every lane reads the same entry) */
        const double theta = ( S[j*ld+j] - S[i*ld+i] )/( 2.0*apq );
        const double t = ( theta >= 0 ? 1.0 : -1.0 )/( fabs( theta ) + sqrt( theta*theta + 1.0 ) );
        const double c = 1.0/sqrt( t*t + 1.0 ), sn = t*c;
        SYNC();
        if( lane < r ){
          const double kp = S[lane*ld+i], kq = S[lane*ld+j];
          S[lane*ld+i] = c*kp - sn*kq; S[lane*ld+j] = sn*kp + c*kq;
        }
        SYNC();
        if( lane < r ){
          const double kp = S[i*ld+lane], kq = S[j*ld+lane];
          S[i*ld+lane] = c*kp - sn*kq; S[j*ld+lane] = sn*kp + c*kq;
          const double vp = EV[lane*ld+i], vq = EV[lane*ld+j];
          EV[lane*ld+i] = c*vp - sn*vq; EV[lane*ld+j] = sn*vp + c*vq;
        }
        SYNC();
      }
  }
  SYNC();
  const double wmax = -WMIN( lane < r ? -fabs( S[lane*ld+lane] ) : 0.0 );
  if( lane < r ) vec[lane] = rhs_i;
  SYNC();
  double y = 0;
  if( lane < r ){
    double s = 0;
    for( int k=0; k<r; k++ ) s += EV[k*ld+lane]*vec[k];
    y = fabs( S[lane*ld+lane] ) > 1e-12*wmax ? s/S[lane*ld+lane] : 0.0;
  }
  SYNC();
  if( lane < r ) vec[lane] = y;
  SYNC();
  double x = 0;
  if( lane < r ) for( int i=0; i<r; i++ ) x += EV[lane*ld+i]*vec[i];
  SYNC();
  return x;
}

/* rkFDQPSolveASM (reference src/rkfd_opt_qp.c:43-181) for the Volume plugin: min x'Qx/2 + c'x s.t. G x >= 0, G's row
 * lane touching the six unknowns of its pair; start point init (lane = unknown).  In: L.VQL = Q (packed lower triangle),
 * cv; out: ans (L.VQV + 2n).  lane = constraint as well (mc <= 64). */
/* With at most RKFD_VQ_NQ = 24 unknowns (four pairs in contact at once) the factor of Q lives in registers as in the Vert QP
 * (rkfd_dev_vertqp.h: rkfdQpFactorT - same arithmetic per entry, same bits): the factorisation, z, the columns of W two per pass
 * with lane = row, the back substitution and the objective without a trip through LDS per term. */
#ifndef RKFD_VQ_NQ
#  if defined(RKFD_SPEC)
#    define RKFD_VQ_NQ ( ( RKFD_SPEC_VOL_NP > 0 && 6*RKFD_SPEC_VOL_NP <= 24 ) ? 6*RKFD_SPEC_VOL_NP : 0 )
#  else
#    define RKFD_VQ_NQ 24
#  endif
#endif
RKFD_DEV void rkfd_vol_qp(const rkfdLds &L, int n, int mc, const double *g, int gp, double init, int nmax)
{
  const int lane = LANE();
  const bool reg = RKFD_VQ_NQ > 0 && nmax <= RKFD_VQ_NQ;
  rkfdQpFactorT<RKFD_VQ_NQ> F;
  F.rd = 0.0;
  const int ldw = mc > 0 ? mc : 1;
  double *Q = L.VQL, *W = L.VQW, *S = L.VS, *EV = L.VEV;
  double *cv = L.VQV, *zv = L.VQV + n, *ans = L.VQV + 2*n, *xv = L.VQV + 3*n, *dv = L.VQV + 4*n;
  double *lam = L.VQV + 5*n;
  const bool onc = lane < mc;
  if( lane < n ) ans[lane] = init;
  SYNC();
  if( reg ){ rkfd_qreg_chol<RKFD_VQ_NQ>( Q, n ); SYNC(); rkfd_qreg_load( F, Q, n ); }
  else rkfd_w_chol<true>( Q, 0, n );
  {
    const double ci = lane < n ? cv[lane] : 0.0;
    const double zi = reg ? rkfd_qreg_fwd( F, n, ci ) : rkfd_w_fwd<true>( Q, 0, n, ci );
    if( lane < n ) zv[lane] = zi;
  }
  SYNC();
#define RKFD_VOL_COND(x) ( g[0]*(x)[6*gp] + g[1]*(x)[6*gp+1] + g[2]*(x)[6*gp+2] + g[3]*(x)[6*gp+3] + g[4]*(x)[6*gp+4] + g[5]*(x)[6*gp+5] )
  int act = 0;
  if( onc ) act = fabs( RKFD_VOL_COND( ans ) - 0.0 ) < RKFD_DEV_TOL;
  unsigned long long hmask = 0; double hobj = 0; int nhist = 0;
  int fail = 0;
  for( int iter=0; ; iter++ ){
    if( iter >= RKFD_QP_MAXITER ){ fail = 1; break; }
    const unsigned long long mask = BALLOT( act );
    const int r = __builtin_popcountll( mask );
    const int rho = __builtin_popcountll( mask & ( lane == 0 ? 0ull : ( ~0ull >> ( 64-lane ) ) ) );
    /* W = L^-1 G' (column rho = this lane's active row), S = W'W, rhs = W'z */
    if( reg ){
      /* lane = row; the active rows in ascending lane order (their columns rho = 0, 1, ...), two per pass */
      unsigned long long mk = mask;
      for( int a0=0; a0<r; a0+=2 ){
        const int l0 = __builtin_ctzll( mk ); mk &= mk - 1ull;
        int l1 = l0;
        if( a0+1 < r ){ l1 = __builtin_ctzll( mk ); mk &= mk - 1ull; }
        const int p0 = 6*BCASTI( gp, l0 ), p1 = 6*BCASTI( gp, l1 );
        double y0 = 0, y1 = 0;
#pragma unroll
        for( int k=0; k<6; k++ ){
          const double g0k = BCAST( g[k], l0 ), g1k = BCAST( g[k], l1 );
          if( lane == p0+k ) y0 = g0k;
          if( lane == p1+k ) y1 = g1k;
        }
        rkfd_qreg_fwd2( F, n, p0 < p1 ? p0 : p1, y0, y1 );
        if( lane < n ){
          W[lane*ldw+a0] = lane < p0 ? 0.0 : y0;
          if( l1 != l0 ) W[lane*ldw+a0+1] = lane < p1 ? 0.0 : y1;
        }
      }
    } else if( onc && act ){
      for( int i=0; i<n; i++ ){
        const int k = i - 6*gp;
        double sacc = ( k >= 0 && k < 6 ) ? ( k == 0 ? g[0] : ( k == 1 ? g[1] : ( k == 2 ? g[2] : ( k == 3 ? g[3] : ( k == 4 ? g[4] : g[5] ) ) ) ) ) : 0.0;
        for( int j=6*gp; j<i; j++ ) sacc -= Q[RKFD_QI( i, j )]*W[j*ldw+rho];
        W[i*ldw+rho] = i < 6*gp ? 0.0 : sacc*Q[RKFD_QI( i, i )];
      }
    }
    SYNC();
    for( int t0=0; t0<r*r; t0+=RKFD_WAVE ){
      const int t = t0 + lane, a = r > 0 ? t/r : 0, b = t - a*r;
      if( t < r*r && b <= a ){
        double sacc = 0;
        for( int i=0; i<n; i++ ) sacc = fma( W[i*ldw+a], W[i*ldw+b], sacc );
        S[a*ldw+b] = sacc; S[b*ldw+a] = sacc;
      }
    }
    double rl = 0;
    if( lane < r ) for( int i=0; i<n; i++ ) rl = fma( W[i*ldw+lane], zv[i], rl );
    SYNC();
    {
      const double l = rkfd_vol_pinv( S, EV, lam, ldw, r, rl );
      if( lane < r ) lam[lane] = l;
    }
    SYNC();
    {
      double ti = 0;
      if( lane < n ){
        for( int a=0; a<r; a++ ) ti = fma( W[lane*ldw+a], lam[a], ti );
        ti -= zv[lane];
      }
      const double xi = reg ? rkfd_qreg_back( F, n, ti ) : rkfd_w_back<true>( Q, 0, n, ti );
      if( lane < n ) xv[lane] = xi;
    }
    SYNC();
    const bool moved = BALLOT( lane < n && !( fabs( xv[lane] - ans[lane] ) < RKFD_DEV_TOL ) ) != 0ull;
    if( !moved ){
      if( lane < n ) ans[lane] = xv[lane];
      const double y = ( onc && act ) ? lam[rho] : 0.0;
      SYNC();
      if( BALLOT( onc && act && y < 0 ) == 0ull ) break;
      const double ymin = WMIN( ( onc && act ) ? y : HUGE_VAL );
      if( onc && act && fabs( y - ymin ) < RKFD_QP_ASM_TOL ) act = 0;
      continue;
    }
    if( lane < n ) dv[lane] = xv[lane] - ans[lane];
    SYNC();
    double tq = HUGE_VAL;
    if( onc && !act ){
      const double gd = RKFD_VOL_COND( dv );
      if( gd < 0 ) tq = ( 0.0 - RKFD_VOL_COND( ans ) )/gd;
    }
    double tmin = WMIN( tq );
    if( !( tmin < 1.0 ) ) tmin = 1.0;
    if( lane < n ) ans[lane] += tmin*dv[lane];
    SYNC();
    if( onc && !act && fabs( RKFD_VOL_COND( ans ) - 0.0 ) < RKFD_DEV_TOL ) act = 1;
    double part = 0;
    if( reg ){
      const double ai = lane < n ? ans[lane] : 0.0;
      const double u = rkfd_qreg_ltv( F, n, ai );
      if( lane < n ) part = 0.5*u*u + cv[lane]*ai;
    } else if( lane < n ){
      double u = ans[lane]/Q[RKFD_QI( lane, lane )];
      for( int j=lane+1; j<n; j++ ) u = fma( Q[RKFD_QI( j, lane )], ans[j], u );
      part = 0.5*u*u + cv[lane]*ans[lane];
    }
    const double objv = WSUM( part );
    const unsigned long long nmask = BALLOT( act );
    const bool seen = lane < nhist && hmask == nmask && !( fabs( hobj/objv - 1.0 ) > RKFD_QP_ASM_TOL );
    if( BALLOT( seen ) != 0ull ) break;
    if( nhist >= RKFD_WAVE ){ fail = 1; break; }
    if( lane == nhist ){ hmask = nmask; hobj = objv; }
    nhist++;
  }
#undef RKFD_VOL_COND
  if( BALLOT( lane < n && !( ans[lane] == ans[lane] ) ) != 0ull ) fail = 1;
  if( fail && lane == 0 ) L.cnt[CNT_QPF] = 1;
  SYNC();
}

/* ------------------------------------------------------------------------ */
/* zLPSolveSimplex / zLPFeasibleBase as restated in the oracle (vol_lp): min c'x s.t. Ax = b, x >= 0; two-phase tableau,
 * entering column by the most negative reduced cost (Bland's rule after 64 pivots).  The tableau lives in REGISTERS: lane =
 * column (columns lane, lane + 64, lane + 128: n + mr + 1 <= 192), six rows and the reduced cost per column; the pivot column and
 * the right-hand side reach every lane through v_readlane, the ratio test is scalar, a pivot is a few dozen vector
 * instructions - no LDS traffic and no barrier inside the iteration.  A (mr x n, row-major), b (mr), c (n; phase 2 only when
 * has_c) are read from LDS once; the result x (n) is written there. */
#define RKFD_LP_EPS 1e-10
#define RKFD_LP_SEL(t, i, h) ( (h) == 0 ? t[0][i] : ( ( RKFD_LP_NS < 3 || (h) == 1 ) ? t[1][i] : t[RKFD_LP_NS-1][i] ) )
#define RKFD_LP_SELC(cc, h) ( (h) == 0 ? cc[0] : ( ( RKFD_LP_NS < 3 || (h) == 1 ) ? cc[1] : cc[RKFD_LP_NS-1] ) )
/* RKFD_LP_NS = column slots per lane: 2 (up to 128 columns: every pair of boxes) or 3 (up to 192: round shapes with up to 16
 * contact-plane conditions) */
template<int RKFD_LP_NS> RKFD_DEV int rkfd_vol_lp_n(const double *A, const double *b, const double *c, double *x, int mr, int n, bool has_c)
{
  const int lane = LANE();
  const int nt = n + mr;                       /* column of the right-hand side */
  const int rl = nt & 63, rh = nt >> 6;
  double T[RKFD_LP_NS][6], cc[RKFD_LP_NS];
  int bas[6];
  double scale = 0;
#pragma unroll
  for( int i=0; i<6; i++ ){
    bas[i] = n+i;
    const double bi = i < mr ? b[i] : 0.0, sg = bi < 0 ? -1.0 : 1.0;
#pragma unroll
    for( int h=0; h<RKFD_LP_NS; h++ ){
      const int j = lane + 64*h;
      T[h][i] = i < mr ? ( j < n ? sg*A[n*i+j] : ( j < nt ? ( j-n == i ? 1.0 : 0.0 ) : ( j == nt ? sg*bi : 0.0 ) ) ) : 0.0;
    }
    if( fabs( bi ) > scale ) scale = fabs( bi );
  }
  /* one loop, one copy of the pivot code.  mode 1: iterations of phase 1 (cost = sum of the artificials); mode 2: artificials left
   * in the base at zero are pivoted out on the first structural column with an entry in their row; mode 3: iterations of phase 2 */
  int ok = 1, mode = 1, it = 0, dstart = 0;
  bool newcost = true;
  for(;;){
    if( newcost ){
      /* reduced costs of the phase that starts */
#pragma unroll
      for( int h=0; h<RKFD_LP_NS; h++ ){
        const int j = lane + 64*h;
        double r = mode == 1 ? ( j >= n && j < nt ? 1.0 : 0.0 ) : ( j < n ? c[j] : 0.0 );
#pragma unroll
        for( int i=0; i<6; i++ )
          if( i < mr ){
            const double cb = mode == 1 ? ( bas[i] >= n ? 1.0 : 0.0 ) : ( bas[i] < n ? c[bas[i]] : 0.0 );
            r -= cb*T[h][i];
          }
        cc[h] = r;
      }
      newcost = false; it = 0;
    }
    int col = -1, row = -1;
    if( mode != 2 ){
      if( it >= 10000 ){ ok = 0; break; }
      const int ncol = mode == 1 ? nt : n;
      /* entering column: the most negative reduced cost, lowest index among equals; Bland's rule after 64 pivots (as the oracle) */
      bool v[RKFD_LP_NS];
      double lm = 0;
#pragma unroll
      for( int h=0; h<RKFD_LP_NS; h++ ){ v[h] = lane + 64*h < ncol && cc[h] < -RKFD_LP_EPS; if( v[h] ) lm = fmin( lm, cc[h] ); }
      double cm = 0;
      if( it < 64 ) cm = WMIN( lm );
#pragma unroll
      for( int h=0; h<RKFD_LP_NS; h++ )
        if( col < 0 && 64*h < ncol ){
          const unsigned long long mk = BALLOT( v[h] && ( it >= 64 || cc[h] == cm ) );
          if( mk ) col = 64*h + __builtin_ctzll( mk );
        }
      if( col < 0 ){
        if( mode == 3 ) break;                      /* optimal */
        /* end of phase 1: feasible when the artificials are (numerically) zero */
        double art = 0;
#pragma unroll
        for( int i=0; i<6; i++ ){
          const double ri = BCAST( RKFD_LP_SEL( T, i, rh ), rl );
          if( i < mr && bas[i] >= n ) art += ri;
        }
        if( art > 1e-9*( 1.0 + scale ) ){ ok = 0; break; }
        mode = 2; dstart = 0;
        continue;
      }
    } else {
#pragma unroll
      for( int i=0; i<6; i++ ) if( row < 0 && i >= dstart && i < mr && bas[i] >= n ) row = i;
      if( row < 0 ){
        if( !has_c ) break;
        mode = 3; newcost = true;
        continue;
      }
      dstart = row + 1;
#pragma unroll
      for( int h=0; h<RKFD_LP_NS; h++ )
        if( col < 0 && 64*h < n ){
          double tr = 0;
#pragma unroll
          for( int i=0; i<6; i++ ) if( i == row ) tr = T[h][i];
          const unsigned long long mk = BALLOT( lane + 64*h < n && fabs( tr ) > 1e-9 );
          if( mk ) col = 64*h + __builtin_ctzll( mk );
        }
      if( col < 0 ) continue;                        /* a redundant row: the artificial stays, at zero */
    }
    const int cl = col & 63, ch = col >> 6;
    double pc[6], rr[6];
#pragma unroll
    for( int i=0; i<6; i++ ){
      pc[i] = BCAST( RKFD_LP_SEL( T, i, ch ), cl );
      rr[i] = BCAST( RKFD_LP_SEL( T, i, rh ), rl );
    }
    if( mode != 2 ){
      /* leaving row: minimum ratio, lowest basic index among equals */
      int brow = 0; double best = 0;
#pragma unroll
      for( int i=0; i<6; i++ )
        if( i < mr && pc[i] > RKFD_LP_EPS ){
          const double r = rr[i]*RKFD_RCP( pc[i] );
          if( row < 0 || r < best - 1e-15 || ( !( r > best + 1e-15 ) && bas[i] < brow ) ){ row = i; best = r; brow = bas[i]; }
        }
      if( row < 0 ){ ok = 0; break; }               /* unbounded */
    }
    {
      double pv = 0;
#pragma unroll
      for( int i=0; i<6; i++ ) if( i == row ) pv = RKFD_RCP( pc[i] );
      const double fc = BCAST( RKFD_LP_SELC( cc, ch ), cl );
#pragma unroll
      for( int h=0; h<RKFD_LP_NS; h++ ){
        double t = 0;
#pragma unroll
        for( int i=0; i<6; i++ ) if( i == row ) t = T[h][i];
        t *= pv;
#pragma unroll
        for( int i=0; i<6; i++ ){
          if( i == row ) T[h][i] = t;
          else if( i < mr && pc[i] != 0.0 ) T[h][i] -= pc[i]*t;
        }
        cc[h] -= fc*t;
      }
#pragma unroll
      for( int i=0; i<6; i++ ) if( i == row ) bas[i] = col;
    }
    it++;
  }
  if( ok ){
    for( int j=lane; j<n; j+=RKFD_WAVE ) x[j] = 0;
    SYNC();
#pragma unroll
    for( int i=0; i<6; i++ ){
      const double ri = BCAST( RKFD_LP_SEL( T, i, rh ), rl );
      if( i < mr && bas[i] < n && lane == 0 ) x[bas[i]] = ri;
    }
  }
  SYNC();
  return ok;
}
RKFD_DEV int rkfd_vol_lp(const double *A, const double *b, const double *c, double *x, int mr, int n, bool has_c, int maxcol)
{
  /* (maxcol: the most columns this world's LPs can have - a property of the world, so the branch is the same for every call) */
  if( maxcol <= 128 ) return rkfd_vol_lp_n<2>( A, b, c, x, mr, n, has_c );
  return rkfd_vol_lp_n<3>( A, b, c, x, mr, n, has_c );
}

/* ------------------------------------------------------------------------ */
/* The friction fix-up of one pair (_rkFDSolverModifyWrenchStatic :677-688, ...Kinetic :830-843) as a small state machine
 * around ONE call site of the simplex (its code is large; inlined at five places it made the kernel 186 KB, three times the
 * instruction cache two CUs share):
 *   stage 1  static: can the wrench be written as forces inside the friction pyramids at the polygon's corners?  yes -> the
 *            wrench stays; no -> stage 2
 *   stage 2  kinetic: the normal force spread over the corners (resultant and centre kept), every corner sliding its own way,
 *            cost = -w / |w| . friction; infeasible -> stage 3
 *   stage 3  "safety": only the resultant is kept (one equality row), the cost gets the reference's extra term (:795-812)
 * w = wrench in the pair frame ( f.axis[0..2], n.axis[0..2] ); a kinetic outcome replaces w[1], w[2], w[3].  Returns 1 when the
 * wrench was replaced (kinetic), 0 when it stays (static).  All lanes call it with the same arguments. */
RKFD_DEV int rkfd_vol_friction(const rkfdDevModel &m, const rkfdLds &L, int k, int ci, double *w, int stage)
{
  const int lane = LANE();
  const int NCP = m.vol_ncp, ncp = L.VI[2*k], P = m.pyramid;
  const double *pl = &L.VPL[8*NCP*k], *vd = &L.VD[RKFD_VD*k];
  const double *ax = vd + RKFD_VD_AX;
  const int PN = P*NCP;
  double *A = L.VLP, *b = A + 6*PN, *c = b + 6, *x = c + PN;      /* layout of L.VLP: A [6 x PN], b [6], c [PN], x [PN] */
  double r0 = 0, r1 = 0, sx = 0, sy = 0;                           /* lane = corner (kinetic stages) */
  if( lane < ncp ){
    const double *c2 = &pl[8*lane];
    r0 = d_dot( c2, ax+3 ); r1 = d_dot( c2, ax+6 );
    /* _rkFDSolverPlaneVertSlideDir (:759-776) */
    const double t1 = c2[6], t2 = c2[7], nv = sqrt( t1*t1 + t2*t2 );
    if( !( fabs( nv ) < RKFD_DEV_TOL ) ){
      const double ww = ( 1.0 - exp( -1.0*m.fric_w*nv ) )*m.ci_kf[ci]/nv;
      sx = -ww*t1; sy = -ww*t2;
    }
  }
  for(;;){
    int mr, n;
    if( stage == 1 ){
      /* _rkFDSolverModifyWrenchStaticConstraint (:652-675) */
      const double mu = m.ci_sf[ci];
      mr = 6; n = P*ncp;
      for( int j=lane; j<n; j+=RKFD_WAVE ){
        const int kk = j/P, i = j - kk*P;
        const double *c2 = &pl[8*kk];
        const double q0 = d_dot( c2, ax+3 ), q1 = d_dot( c2, ax+6 );
        const double PI = 3.14159265358979323846;
        double th = 0.0, sn, cs;
        for( int q=0; q<i; q++ ) th += 2.0*PI/P;
        d_sincos( th + 0.0, &sn, &cs );
        const double a1 = q1, a2 = -q0, a3 = mu*cs, a4 = mu*sn;
        A[j] = 1.0; A[n+j] = a1; A[2*n+j] = a2; A[3*n+j] = a3; A[4*n+j] = a4; A[5*n+j] = -( a2*a4 + a1*a3 );
      }
      if( lane == 0 ){ b[0] = w[0]; b[1] = w[4]; b[2] = w[5]; b[3] = w[1]; b[4] = w[2]; b[5] = w[3]; }
    } else if( stage == 2 ){
      /* _rkFDSolverModifyWrenchKineticConstraint / ...EvalFunc (:743-793) */
      mr = 3; n = ncp;
      if( lane < n ){
        A[lane] = 1.0; A[n+lane] = r1; A[2*n+lane] = -r0;
        double wn[3];
#pragma unroll
        for( int i=0; i<3; i++ ) wn[i] = fabs( w[i+1] ) < RKFD_DEV_TOL ? 0.0 : 1.0/w[i+1];
        c[lane] = -wn[0]*sx - wn[1]*sy - wn[2]*( r0*sy - r1*sx );
      }
      if( lane == 0 ){ b[0] = w[0]; b[1] = w[4]; b[2] = w[5]; }
    } else {
      /* _rkFDSolverModifyWrenchKineticEvalFuncSafety (:795-812): the first row of A and b, the cost of stage 2 plus a term */
      mr = 1; n = ncp;
      if( lane < n ){
        double wn[2];
#pragma unroll
        for( int i=0; i<2; i++ ) wn[i] = fabs( w[i+3] ) < RKFD_DEV_TOL ? 0.0 : 1.0/w[i+3];
        c[lane] += wn[0]*r0 - wn[1]*r1;
      }
    }
    if( lane < n && stage != 1 ) x[lane] = 0;
    SYNC();
    const int ok = rkfd_vol_lp( A, b, c, x, mr, n, stage != 1, PN + 7 );
    if( stage == 1 ){ if( ok ) return 0; stage = 2; continue; }
    if( stage == 2 && !ok ){ stage = 3; continue; }
    break;
  }
  /* _rkFDSolverModifyWrenchKineticTotalWrench (:814-828): lane order = list order */
  double *red = A;       /* (the constraint matrix is dead) */
  if( lane < ncp ){ const double fx = sx*x[lane], fy = sy*x[lane]; red[3*lane] = fx; red[3*lane+1] = fy; red[3*lane+2] = r0*fy - r1*fx; }
  SYNC();
  double a1 = 0, a2 = 0, a3 = 0;
  for( int i=0; i<ncp; i++ ){ a1 += red[3*i]; a2 += red[3*i+1]; a3 += red[3*i+2]; }
  w[1] = a1; w[2] = a2; w[3] = a3;
  SYNC();
  return 1;
}

/* ------------------------------------------------------------------------ */
/* the rigid branch proper (_rkFDSolverVolume, :939-957).  Preconditions as for rkfd_phase_mlcp. */
template<bool prof> RKFD_DEV void rkfd_phase_volume(const rkfdDevModel &m, const rkfdLds &L, bool doUpRef, unsigned long long *pc)
{
  unsigned long long q0 = prof ? RKFD_CLOCK() : 0ull, q1;
#define VST(k) do{ if( prof ){ q1 = RKFD_CLOCK(); pc[k] += q1 - q0; q0 = q1; } }while(0)
  const int lane = LANE();
  const int np = L.cnt[CNT_NVP], n = 6*np, ld = n+1;
  const int NCP = m.vol_ncp;
  const int NLV = m.nlevel, NL = m.nlink, NR = m.npurow, NSD = m.nside;
  const unsigned char *TOP = L.PL + NL*NLV, *FSL = TOP + NL, *FLK = FSL + NL;
  const double dt = m.dt;

  /* b = dt x free 6-D relative acceleration at the centre + relative velocity (:143-154, :214-226); the moving sides */
  if( lane < np ){
    const double *vd = &L.VD[RKFD_VD*lane];
    const int *rec = &RELOAD( m.vol_pair )[8*L.VI[2*lane+1]];
    const int la = RKFD_VP_LA( rec ), lb = RKFD_VP_LB( rec );
    const double x[3] = { vd[RKFD_VD_C], vd[RKFD_VD_C+1], vd[RKFD_VD_C+2] };
    double ta[3], tb[3];
    d_cross( &L.AC[6*la], x, ta ); d_cross( &L.AC[6*lb], x, tb );
#pragma unroll
    for( int k=0; k<3; k++ ){
      const double al = ( L.AC[6*la+3+k] + ta[k] ) - ( L.AC[6*lb+3+k] + tb[k] ) + vd[RKFD_VD_CA+k];
      const double aa = L.AC[6*la+k] - L.AC[6*lb+k];
      L.MB[6*lane+k] = al*dt + vd[RKFD_VD_VEL+k];
      L.MB[6*lane+3+k] = aa*dt + vd[RKFD_VD_VEL+3+k];
    }
    if( NSD == 1 ) L.tgt[lane] = 0;
#pragma unroll
    for( int sd=0; sd<2; sd++ ){
      const int a = sd == 0 ? la : lb;
      const int top = TOP[a];
      if( NSD == 1 && top == 255 ) continue;
      const int lit = L.LI[top == 255 ? 0 : top], jtt = RKFD_LI_JT( lit );
      const int d0 = RKFD_LI_DEPTH( lit ) + ( RKFD_JT_IS1( jtt ) ? 0 : 1 );
      const unsigned e = (unsigned)a | ( (unsigned)RKFD_LI_DEPTH( L.LI[a] ) << 8 ) | ( (unsigned)top << 14 ) | ( (unsigned)d0 << 22 )
                       | ( jtt == RKFD_JOINT_FLOAT ? 1u << 29 : 0u ) | ( (unsigned)sd << 30 ) | ( top != 255 ? 1u << 31 : 0u );
      L.tgt[NSD == 1 ? lane : 2*lane+sd] = (int)e;
    }
  }
  if( lane < NL ){
    const int jt = RKFD_LI_JT( L.LI[lane] );
    if( RKFD_JT_IS1( jt ) ) L.MS[3*lane+2] = sqrt( L.MS[3*lane+0] );
  }
  if( m.has_brf ) rkfd_brf_before_probes( m, L );
  SYNC();
  VST(14);
  /* probes (reference :176-211): lane = column 6 c + i, a unit world force (i < 3) or torque at the centre of pair c, + on
   * cell[0], - on cell[1]; the walk is the one of the MLCP phase */
  if( lane < n ){
    const int col = lane, c = col/6, ia = col - 6*c;
    const double *vd = &L.VD[RKFD_VD*c];
    double W[6] = {0,0,0,0,0,0};
    {
      const double x[3] = { vd[RKFD_VD_C], vd[RKFD_VD_C+1], vd[RKFD_VD_C+2] };
      const int ix = ia < 3 ? ia : ia-3;
      const double ax[3] = { ix == 0 ? 1.0 : 0.0, ix == 1 ? 1.0 : 0.0, ix == 2 ? 1.0 : 0.0 };
      if( ia < 3 ){ d_cross( x, ax, W ); W[3] = ax[0]; W[4] = ax[1]; W[5] = ax[2]; }
      else { W[0] = ax[0]; W[1] = ax[1]; W[2] = ax[2]; }
    }
    for( int s2=0; s2<NSD; s2++ ){
      const unsigned e = (unsigned)L.tgt[c*NSD+s2];
      if( !RKFD_CS_VALID( e ) ) continue;
      const int a = RKFD_CS_LINK( e ), da = RKFD_CS_DEPTH( e ), d0 = RKFD_CS_D0( e );
      double dp[6];
      const double sg = RKFD_CS_SIDE( e ) == 0 ? -1.0 : 1.0;
#pragma unroll
      for( int k=0; k<6; k++ ) dp[k] = sg*W[k];
      double *pu = &L.PU[RKFD_PU_AT( m, s2, col, 0 )];
      const unsigned char *path = &L.PL[a*NLV];
      for( int d=da; d>=d0; d-- ){
        const int il = path[d];
        double du = 0;
#pragma unroll
        for( int k=0; k<6; k++ ) du += L.S[6*il+k]*dp[k];
        du = -du;
        pu[d] = du*L.MS[3*il+2];
        const double t = du*L.MS[3*il+0];
#pragma unroll
        for( int k=0; k<6; k++ ) dp[k] = fma( L.U[6*il+k], t, dp[k] );
      }
      if( RKFD_CS_FLOAT( e ) ){
        double rhs[6], y[6], Lr[21];
#pragma unroll
        for( int k=0; k<6; k++ ) rhs[k] = -dp[k];
        d_chol6_load( &L.CHOL[21*FSL[RKFD_CS_TOP( e )]], Lr );
        d_chol6_fwd( Lr, rhs, y );
#pragma unroll
        for( int k=0; k<6; k++ ) pu[NLV+k] = y[k];
      }
    }
  }
  SYNC();
  VST(15);
  /* A(r,k) = sum over the joints common to both paths of nu_r nu_k (no relaxation here: it enters the QP) */
  for( int e0=0; e0<n*n; e0+=RKFD_WAVE ){
    const int e = e0 + lane;
    if( e < n*n ){
      const int r = e/n, k = e - r*n, cr = r/6, ck = k/6;
      double acc = 0;
      for( int sr=0; sr<NSD; sr++ ) for( int sk=0; sk<NSD; sk++ ){
        const unsigned er = (unsigned)L.tgt[cr*NSD+sr], ek = (unsigned)L.tgt[ck*NSD+sk];
        if( !RKFD_CS_VALID( er ) || !RKFD_CS_VALID( ek ) || RKFD_CS_TOP( er ) != RKFD_CS_TOP( ek ) ) continue;
        const double *pr = &L.PU[RKFD_PU_AT( m, sr, r, 0 )], *pk = &L.PU[RKFD_PU_AT( m, sk, k, 0 )];
        const int a = RKFD_CS_LINK( er ), b = RKFD_CS_LINK( ek ), d0 = RKFD_CS_D0( er );
        int dc = RKFD_CS_DEPTH( er ) < RKFD_CS_DEPTH( ek ) ? RKFD_CS_DEPTH( er ) : RKFD_CS_DEPTH( ek );
        if( a != b ){
          int d = d0;
          while( d <= dc && L.PL[a*NLV+d] == L.PL[b*NLV+d] ) d++;
          dc = d-1;
        }
        for( int d=d0; d<=dc; d++ ) acc = fma( pr[d], pk[d], acc );
        if( RKFD_CS_FLOAT( er ) )
          for( int q=0; q<6; q++ ) acc = fma( pr[NLV+q], pk[NLV+q], acc );
      }
      L.MA[r*ld+k] = acc;
    }
  }
  SYNC();
  VST(6);
  /* _rkFDSolverQPCreate (:496-528): Q = sum_p A_p' qv A_p + l, c = sum_p A_p' ( qv b_p + cv ); qv from the accumulated
   * integrals: [ s 1, -[pc x] ; [pc x], -mm ] */
  double *cvq = L.VQV;
  for( int e0=0; e0<n*( n+1 ); e0+=RKFD_WAVE ){
    const int e = e0 + lane;
    if( e < n*( n+1 ) ){
      const int r = e/( n+1 ), k = e - r*( n+1 );      /* k == n: the linear term */
      if( k <= r || k == n ){
        double acc = 0;
        for( int p=0; p<np; p++ ){
          const double *ac = &L.VD[RKFD_VD*p + RKFD_VD_ACC];
          const double s = ac[0], pc[3] = { ac[1], ac[2], ac[3] };
          /* t = qv y, y = A_p[:,k] (or b_p), then acc += A_p[:,r] . t */
          double y[6], t[6];
#pragma unroll
          for( int i=0; i<6; i++ ) y[i] = k == n ? L.MB[6*p+i] : L.MA[( 6*p+i )*ld+k];
          double x1[3], x2[3];
          d_cross( pc, y+3, x1 ); d_cross( pc, y, x2 );
          t[0] = s*y[0] - x1[0]; t[1] = s*y[1] - x1[1]; t[2] = s*y[2] - x1[2];
          t[3] = x2[0] - ( ac[4]*y[3] + ac[5]*y[4] + ac[6]*y[5] );
          t[4] = x2[1] - ( ac[5]*y[3] + ac[7]*y[4] + ac[8]*y[5] );
          t[5] = x2[2] - ( ac[6]*y[3] + ac[8]*y[4] + ac[9]*y[5] );
          if( k == n ){
#pragma unroll
            for( int i=0; i<6; i++ ) t[i] += ac[10+i];
          }
#pragma unroll
          for( int i=0; i<6; i++ ) acc = fma( L.MA[( 6*p+i )*ld+r], t[i], acc );
        }
        if( k == n ) cvq[r] = acc;
        else {
          if( k == r ) acc += m.ci_l[RKFD_VP_CI( &RELOAD( m.vol_pair )[8*L.VI[2*( r/6 )+1]] )];
          L.VQL[RKFD_QI( r, k )] = acc;
        }
      }
    }
  }
  /* _rkFDSolverFrictionConstraint (:121-138): lane = constraint, pair by pair: the normal row, then one row per condition */
  int mc = 0, gp = 0, gk = -1;
  for( int p=0; p<np; p++ ){
    const int ncp = L.VI[2*p];
    if( lane >= mc && lane < mc+1+ncp ){ gp = p; gk = lane - mc - 1; }
    mc += 1 + ncp;
  }
  double g[6] = {0,0,0,0,0,0};
  if( lane < mc ){
    const double *ax = &L.VD[RKFD_VD*gp + RKFD_VD_AX];
    if( gk < 0 ){ g[0] = ax[0]; g[1] = ax[1]; g[2] = ax[2]; }
    else {
      const double *c2 = &L.VPL[8*( NCP*gp + gk )];
      const double nv = -d_dot( c2+3, c2 ), n2 = d_dot( c2+3, ax+6 ), n1 = -d_dot( c2+3, ax+3 );
      g[0] = nv*ax[0]; g[1] = nv*ax[1]; g[2] = nv*ax[2];
      g[3] = n2*ax[3]; g[4] = n2*ax[4]; g[5] = n2*ax[5];
      g[3] += n1*ax[6]; g[4] += n1*ax[7]; g[5] += n1*ax[8];
    }
  }
  /* start point (_rkFDSolverQPInit, :530-542): a unit normal force per pair */
  double init = 0;
  if( lane < n ){ const int c = lane/6, i = lane - 6*c; init = i < 3 ? L.VD[RKFD_VD*c + RKFD_VD_AX + i] : 0.0; }
  SYNC();
  VST(24);
  rkfd_vol_qp( L, n, mc, g, gp, init, 6*m.vol_np );
  VST(25);
  /* _rkFDSolverQP (:547), _rkFDSolverSetForce (:552-568; the offset stays behind a pair without conditions, as in the reference) */
  {
    const double *ans = L.VQV + 2*n;
    if( lane < np ){
      int off = 0;
      for( int p=0; p<lane; p++ ) if( L.VI[2*p] != 0 ) off += 6;
      double *vd = &L.VD[RKFD_VD*lane];
      double wv[6] = {0,0,0,0,0,0};
      if( L.VI[2*lane] != 0 ){
#pragma unroll
        for( int k=0; k<6; k++ ) wv[k] = ans[off+k]/dt;
        const double *ax = vd + RKFD_VD_AX;
        if( ( fabs( wv[0] ) < RKFD_DEV_TOL && fabs( wv[1] ) < RKFD_DEV_TOL && fabs( wv[2] ) < RKFD_DEV_TOL ) || d_dot( wv, ax ) < RKFD_DEV_TOL ){
#pragma unroll
          for( int k=0; k<6; k++ ) wv[k] = 0;
        }
      }
      /* _rkFDSolverModifyNormalForceCenter (:580-631) */
      const double *ax = vd + RKFD_VD_AX;
      const int nn = L.VI[2*lane];
      const double fn = d_dot( ax, wv );
      if( !( fn < RKFD_DEV_TOL ) && nn >= 1 ){
        const double *pl = &L.VPL[8*NCP*lane];
        double r0[3];
        {
          const double k1 = -d_dot( ax+6, wv+3 )/fn, k2 = d_dot( ax+3, wv+3 )/fn;
          r0[0] = k1*ax[3] + k2*ax[6]; r0[1] = k1*ax[4] + k2*ax[7]; r0[2] = k1*ax[5] + k2*ax[8];
        }
        int flag = 0;
        for( int k=0; k<nn; k++ ){
          const double *c0 = &pl[8*( ( k+3*nn-3 ) % nn )], *c1 = &pl[8*( ( k+3*nn-2 ) % nn )], *c2 = &pl[8*( ( k+3*nn-1 ) % nn )], *c3 = &pl[8*k];
          const double dir[3] = { c2[0]-c1[0], c2[1]-c1[1], c2[2]-c1[2] };
          const double d = d_dot( dir, dir );
          if( fabs( d ) < RKFD_DEV_TOL ) continue;
          double tmp[3] = { r0[0]-c1[0], r0[1]-c1[1], r0[2]-c1[2] };
          if( d_dot( tmp, c1+3 ) > RKFD_DEV_TOL ) continue;
          const double s = d_dot( dir, tmp )/d;
          double r[3];
          int stop = 1;
          if( s < RKFD_DEV_TOL ){
            if( flag ) break;
            tmp[0] = c0[0]-c1[0]+dir[0]; tmp[1] = c0[1]-c1[1]+dir[1]; tmp[2] = c0[2]-c1[2]+dir[2];
            const double q = RKFD_DEV_TOL/sqrt( d_dot( tmp, tmp ) );
            r[0] = c1[0]+q*tmp[0]; r[1] = c1[1]+q*tmp[1]; r[2] = c1[2]+q*tmp[2];
          } else if( s < 1.0-RKFD_DEV_TOL ){
            r[0] = c1[0]+s*dir[0]; r[1] = c1[1]+s*dir[1]; r[2] = c1[2]+s*dir[2];
            r[0] += RKFD_DEV_TOL*c1[3]; r[1] += RKFD_DEV_TOL*c1[4]; r[2] += RKFD_DEV_TOL*c1[5];
          } else {
            tmp[0] = c3[0]-c2[0]-dir[0]; tmp[1] = c3[1]-c2[1]-dir[1]; tmp[2] = c3[2]-c2[2]-dir[2];
            const double q = RKFD_DEV_TOL/sqrt( d_dot( tmp, tmp ) );
            r[0] = c2[0]+q*tmp[0]; r[1] = c2[1]+q*tmp[1]; r[2] = c2[2]+q*tmp[2];
            flag = 1; stop = 0;
          }
          /* _rkFDSolverModifyNormForceCenterTrq (:573-578) */
          const double nt = d_dot( ax, wv+3 ), k1 = fn*d_dot( ax+6, r ), k2 = -fn*d_dot( ax+3, r );
#pragma unroll
          for( int q=0; q<3; q++ ) wv[3+q] = nt*ax[q] + k1*ax[3+q] + k2*ax[6+q];
          if( stop ) break;
        }
      }
#pragma unroll
      for( int k=0; k<6; k++ ) vd[RKFD_VD_W+k] = wv[k];
    }
  }
  SYNC();
  VST(26);
  /* _rkFDSolverModifyWrench (:869-916), pair by pair (the simplex runs on the whole wave) */
  for( int p=0; p<np; p++ ){
    double *vd = &L.VD[RKFD_VD*p];
    const double *ax = vd + RKFD_VD_AX;
    const int ncp = L.VI[2*p], prm = L.VI[2*p+1];
    const int ci = RKFD_VP_CI( &RELOAD( m.vol_pair )[8*prm] );
    if( ncp == 0 ) continue;
    double wr[6];
#pragma unroll
    for( int k=0; k<6; k++ ) wr[k] = vd[RKFD_VD_W+k];
    if( fabs( d_dot( wr, ax ) ) < RKFD_DEV_TOL ) continue;
    double w[6];
#pragma unroll
    for( int i=0; i<3; i++ ){ w[i] = d_dot( wr, ax+3*i ); w[i+3] = d_dot( wr+3, ax+3*i ); }
    const double fn = w[0], fs = sqrt( w[1]*w[1] + w[2]*w[2] ), sf = m.ci_sf[ci];
    double tl = 0;
    for( int k=0; k<ncp; k++ ){
      const double *c2 = &L.VPL[8*( NCP*p + k )];
      const double q0 = d_dot( c2, ax+3 ), q1 = d_dot( c2, ax+6 ), rl = sqrt( q0*q0 + q1*q1 );
      if( tl < rl ) tl = rl;
    }
    int kin = 0, setf = 1;
    if( fabs( tl ) < RKFD_DEV_TOL ){
      w[3] = w[4] = w[5] = 0;
      if( !( fabs( fs ) < RKFD_DEV_TOL ) && fs > sf*fn ){
        /* _rkFDSolverModifyWrenchKineticCenter (:715-731) */
        const double t1 = vd[RKFD_VD_TC], t2 = vd[RKFD_VD_TC+1], nv = sqrt( t1*t1 + t2*t2 );
        if( fabs( nv ) < RKFD_DEV_TOL ){ w[1] = w[2] = 0; }
        else {
          const double t = ( 1.0 - exp( -1.0*m.fric_w*nv ) )*m.ci_kf[ci]*w[0]/nv;
          w[1] = -t*t1; w[2] = -t*t2;
        }
        kin = 1;
      }
    } else {
      /* kinetic straight away when the friction limit or the twisting limit is passed, else the static test first (:899-914) */
      const int stage = ( ( !( fabs( fs ) < RKFD_DEV_TOL ) && fs > sf*fn ) || fabs( w[3] ) > tl*w[0] ) ? 2 : 1;
      kin = rkfd_vol_friction( m, L, p, ci, w, stage );
      setf = kin;
    }
    if( setf && lane < 6 ){
      /* _rkFDSolverModifyWrenchSetForce (:858-867) */
      const int h = lane < 3 ? 0 : 3, q = lane - h;
      vd[RKFD_VD_W+lane] = w[h]*ax[q] + w[h+1]*ax[3+q] + w[h+2]*ax[6+q];
    }
    (void)kin; (void)doUpRef;     /* the pair's stick / slip type (cpd->type, :694,849) is write-only in the reference: not kept */
    SYNC();
  }
  SYNC();
  VST(27);
  /* the forces in the order of the probe columns, then the inputs of the delta sweep as in the MLCP phase */
  if( lane < n ) L.MF[lane] = L.VD[RKFD_VD*( lane/6 ) + RKFD_VD_W + lane%6];
  SYNC();
  {
    const int ntask = NL + 6*m.nfloat;
    for( int t0=0; t0<ntask; t0+=RKFD_WAVE ){
      const int t = t0 + lane;
      const bool isl = t < NL, isf = !isl && t < ntask;
      const int fq = isf ? ( t-NL )%6 : 0;
      const int link = isl ? t : ( isf ? FLK[( t-NL )/6] : 0 );
      const int lii = L.LI[link], jt = RKFD_LI_JT( lii );
      const bool is1 = isl && RKFD_JT_IS1( jt );
      const int dpt = is1 ? RKFD_LI_DEPTH( lii ) : m.pu_d0;
      const int row = isf ? NLV+fq : dpt;
      double sum = 0;
      for( int cs=0; cs<np*NSD; cs++ ){
        const unsigned e = (unsigned)L.tgt[cs];
        const int c = NSD == 1 ? cs : cs >> 1;
        const double *pu = &L.PU[RKFD_PU_AT( m, NSD == 1 ? 0 : ( cs & 1 ), 6*c, row )];
        double v = 0;
#pragma unroll
        for( int k=0; k<6; k++ ) v += L.MF[6*c+k]*pu[k*NR];
        const bool onp = RKFD_CS_VALID( e ) && ( isf ? RKFD_CS_TOP( e ) == link
                       : ( RKFD_CS_DEPTH( e ) >= dpt && RKFD_CS_D0( e ) <= dpt && L.PL[RKFD_CS_LINK( e )*NLV+dpt] == link ) );
        sum += onp ? v : 0.0;
      }
      if( is1 ) L.MS[3*link+1] = sum*L.MS[3*link+2];
      if( isf ) L.U[6*link+fq] = sum;
    }
  }
  SYNC();
  VST(23);
#undef VST
}

/* the break test of the breakable float joints (rkfd_dev_brf.h) under the Volume plugin: the rigid pairs in volumetric contact
 * act through one wrench each - force and torque about the pair's centre, + on cell[0], - on cell[1] (VD_W) */
RKFD_DEV void rkfd_brf_break_test_vol(const rkfdDevModel &m, const rkfdLds &L, bool solved)
{
  const int lane = LANE();
  bool breaks = false;      /* (verdicts are written after every lane has read the states of this evaluation, as in rkfd_brf_break_test) */
  if( lane < m.nlink && L.BRK[lane] == RKFD_BRF_ATTACHED ){
    const double *XF = &L.XF[12*rkfd_brf_fslot( m, L, lane )];
    double w[6];
#pragma unroll
    for( int k=0; k<6; k++ ) w[k] = XF[k];
    if( solved ){
      const int np = L.cnt[CNT_NVP];
      for( int p=0; p<np; p++ ){
        const double *vd = &L.VD[RKFD_VD*p];
        const int *rec = &RELOAD( m.vol_pair )[8*L.VI[2*p+1]];
        const double x[3] = { vd[RKFD_VD_C], vd[RKFD_VD_C+1], vd[RKFD_VD_C+2] }, f[3] = { vd[RKFD_VD_W], vd[RKFD_VD_W+1], vd[RKFD_VD_W+2] };
        double t[3];
        d_cross( x, f, t );
        t[0] += vd[RKFD_VD_W+3]; t[1] += vd[RKFD_VD_W+4]; t[2] += vd[RKFD_VD_W+5];
#pragma unroll
        for( int sd=0; sd<2; sd++ ){
          int k = sd == 0 ? RKFD_VP_LA( rec ) : RKFD_VP_LB( rec );
          while( k != lane && k >= 0 && L.BRK[k] == RKFD_BRF_ATTACHED ) k = RKFD_LI_PAR( L.LI[k] );
          if( k == lane ){
            const double sg = sd == 0 ? -1.0 : 1.0;      /* the bias is minus the external wrench */
            w[0] = fma( sg, t[0], w[0] ); w[1] = fma( sg, t[1], w[1] ); w[2] = fma( sg, t[2], w[2] );
            w[3] = fma( sg, f[0], w[3] ); w[4] = fma( sg, f[1], w[4] ); w[5] = fma( sg, f[2], w[5] );
          }
        }
      }
    }
    breaks = rkfd_brf_decide( m, L, lane, XF, w );
  }
  SYNC();
  if( breaks ) L.BRK[lane] = RKFD_BRF_BROKEN;
  SYNC();
}

#endif /* RKFD_DEV_VOLUME_H */
