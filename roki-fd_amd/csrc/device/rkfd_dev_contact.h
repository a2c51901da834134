/* rkfd_dev_contact.h - part of the device code of the batched rkFDUpdate step (see rkfd_device.h):
 * phases: vertex collision, wrench accumulation, penalty forces, velocity part of the MLCP bias.
 * Included by rkfd_device.h only, in this order; compiles for gfx950 and under the lane emulator. */
#ifndef RKFD_DEV_CONTACT_H
#define RKFD_DEV_CONTACT_H

/* ------------------------------------------------------------------------ */
/* point kinematics in world coordinates from spatial quantities at the origin */
RKFD_DEV void d_point_vel(const double *V, const double *x, double *v)
{
  double t[3];
  d_cross( V, x, t );
  v[0] = V[3]+t[0]; v[1] = V[4]+t[1]; v[2] = V[5]+t[2];
}
RKFD_DEV void d_point_acc(const double *A, const double *V, const double *x, double *a)
{
  double v[3], t[3], s[3];
  d_point_vel( V, x, v );
  d_cross( A, x, t ); d_cross( V, v, s );
  a[0] = A[3]+t[0]+s[0]; a[1] = A[4]+t[1]+s[1]; a[2] = A[5]+t[2]+s[2];
}

/* collision detection for convex shapes, lane = candidate vertex.
 * rkCDColChkVert [RoKi, restated as in oracle/rkfd_oracle.c collision()] + rkFDCDUpdate
 * (reference src/rkfd_cd.c:33-49).  Builds the rigid / elastic contact lists in candidate order. */
RKFD_DEV void rkfd_phase_collision(const rkfdDevModel &m, const rkfdLds &L)
{
  const int lane = LANE();
  const unsigned long long below = lane == 0 ? 0ull : ( ~0ull >> ( 64-lane ) );
  int base_act = 0, base_rg = 0, base_el = 0, ovf = 0;
  /* slots are re-assigned chunk by chunk: with more than one chunk the old anchors are read from a copy */
  const double *oldref = L.REF;
  if( m.ncand > RKFD_WL ){
    for( int k=lane; k<3*m.maxact; k+=RKFD_WL ) L.RTMP[k] = L.REF[k];
    oldref = L.RTMP;
    SYNC();
  }
  /* candidates are swept 64 at a time; slots and list positions keep candidate order */
  for( int c0=0; c0<m.ncand; c0+=RKFD_WL ){
    const bool on = c0+lane < m.ncand;
    const int j = on ? c0+lane : 0;
    int is_act = 0, is_rg = 0, is_el = 0, fbest = -1;
    double x[3] = {0,0,0}, y[3] = {0,0,0}, smax = -HUGE_VAL, RB[9], pB[3], RA[9], pA[3] = {0,0,0};
    const int cinf = L.CIp[j];
#pragma unroll
    for( int k=0; k<9; k++ ){ RB[k] = 0; RA[k] = 0; }
    pB[0] = pB[1] = pB[2] = 0;
    if( on ){
      const int la = RKFD_CI_A( cinf ), lb = RKFD_CI_B( cinf );
      double vl[3], rr[3], bs[4];
#pragma unroll
      for( int k=0; k<6; k++ ){ RA[k] = L.XA[6*la+k]; RB[k] = L.XA[6*lb+k]; }
#pragma unroll
      for( int k=0; k<3; k++ ){ RA[6+k] = L.XB[6*la+k]; RB[6+k] = L.XB[6*lb+k]; }
#pragma unroll
      for( int k=0; k<3; k++ ){ pA[k] = L.XB[6*la+3+k]; pB[k] = L.XB[6*lb+3+k]; vl[k] = RELOAD( m.cand_vert )[3*j+k]; }
#pragma unroll
      for( int k=0; k<4; k++ ) bs[k] = RELOAD( m.cand_bs )[4*j+k];
      d_mulv( RA, vl, x );
      x[0] += pA[0]; x[1] += pA[1]; x[2] += pA[2];
      rr[0] = x[0]-pB[0]; rr[1] = x[1]-pB[1]; rr[2] = x[2]-pB[2];
      d_tmulv( RB, rr, y );
      /* broad phase: a vertex outside the sphere around the other shape cannot touch it (no plane is read for it;
       * the face loop of a chunk runs as long as its nearest candidate needs) */
      const double e0 = y[0]-bs[0], e1 = y[1]-bs[1], e2 = y[2]-bs[2];
      const bool nearb = e0*e0 + e1*e1 + e2*e2 <= bs[3];
      const int f0 = L.CFO[j], nf = nearb ? RKFD_CI_NF( cinf ) : 0;
      for( int f=f0; f<f0+nf; f++ ){
        const double sd = m.planes[4*f]*y[0] + m.planes[4*f+1]*y[1] + m.planes[4*f+2]*y[2] - m.planes[4*f+3];
        if( sd > smax ){ smax = sd; fbest = f; }
      }
      is_act = fbest >= 0 && smax < RKFD_DEV_TOL;
      /* under the Volume plugin a rigid pair goes by its intersection volume (rkfd_dev_volume.h), not by contact vertices */
      if( m.vol_np > 0 && m.ci_type[RKFD_CI_CI( cinf )] == RKFD_CONTACT_RIGID ) is_act = 0;
    }
    /* anchors of the contacts that persist, read at their OLD slots before anything is rewritten */
    double oref[3] = {0,0,0};
    const int was = on ? L.act[j] : 0;
    if( was ){ const int ri = RIDX( j ); oref[0] = oldref[3*ri]; oref[1] = oldref[3*ri+1]; oref[2] = oldref[3*ri+2]; }
    LDS_FENCE();
    /* active contacts get a slot in the per-contact arrays (capacity m.maxact) in candidate order */
    const unsigned long long mact = BALLOT( is_act );
    const int slot = base_act + __builtin_popcountll( mact & below );
    if( is_act && slot >= m.maxact ){ is_act = 0; }
    if( on ){
      if( is_act ){
        const double n[3] = { m.planes[4*fbest], m.planes[4*fbest+1], m.planes[4*fbest+2] };
        const double pro[3] = { y[0]-smax*n[0], y[1]-smax*n[1], y[2]-smax*n[2] };
        double nw[3], t1[3], t2[3], ref[3], rw[3];
        L.asl[j] = slot;
        L.CX[3*slot] = x[0]; L.CX[3*slot+1] = x[1]; L.CX[3*slot+2] = x[2];
        L.PRO[3*slot] = pro[0]; L.PRO[3*slot+1] = pro[1]; L.PRO[3*slot+2] = pro[2];
        d_mulv( RB, n, nw );
        if( !was ){
          L.act[j] = 1; L.typ[j] = RKFD_SF;
          ref[0] = pro[0]; ref[1] = pro[1]; ref[2] = pro[2];
        } else { ref[0] = oref[0]; ref[1] = oref[1]; ref[2] = oref[2]; }
        L.REF[3*slot] = ref[0]; L.REF[3*slot+1] = ref[1]; L.REF[3*slot+2] = ref[2];
        L.FS[slot] = 0;
        d_mulv( RB, ref, rw );
        L.RW[3*slot] = rw[0]+pB[0]; L.RW[3*slot+1] = rw[1]+pB[1]; L.RW[3*slot+2] = rw[2]+pB[2];
        d_ortho_space( nw, t1, t2 );
#pragma unroll
        for( int k=0; k<3; k++ ){ L.AX[6*slot+k] = nw[k]; L.AX[6*slot+3+k] = t1[k]; }
        if( m.has_slide ){
          /* cells in slide mode (rkFDLinkAddSlideVel, rkFDUpdateRefSlide; reference src/rkfd_util.c:26-40,218-237):
           * the surface runs around its slide axis, tangentially to the contact normal.  The frames are at hand
           * only here, so the relative slide velocity and the anchor drift of a committing evaluation are kept per slot. */
          double sv[3] = {0,0,0}, sd[3] = {0,0,0};
#pragma unroll
          for( int k=0; k<2; k++ ){
            const int md = RELOAD( m.cs_mode )[2*j+k];
            if( !md ) continue;
            const double *pr = &RELOAD( m.cs_par )[14*j+7*k];
            const double *Rk = k == 0 ? RA : RB, *pk = k == 0 ? pA : pB;
            const double axl[3] = { pr[1], pr[2], pr[3] }, orl[3] = { pr[4], pr[5], pr[6] };
            double axw[3], orw[3], t[3], s[3];
            d_mulv( Rk, axl, axw ); d_mulv( Rk, orl, orw );
            t[0] = x[0]-pk[0]-orw[0]; t[1] = x[1]-pk[1]-orw[1]; t[2] = x[2]-pk[2]-orw[2];
            d_cross( axw, t, s );
            const double sn = d_dot( s, nw );
            s[0] -= sn*nw[0]; s[1] -= sn*nw[1]; s[2] -= sn*nw[2];
            const double nr = sqrt( d_dot( s, s ) );
            if( fabs( nr ) < RKFD_DEV_TOL ) continue;
            const double g = pr[0]/nr, sg = k == 0 ? 1.0 : -1.0;
            sv[0] += sg*g*s[0]; sv[1] += sg*g*s[1]; sv[2] += sg*g*s[2];
            /* anchor drift: -+ dt slide_vel along s, rotated into the frame the reference's index picks */
            const double h = ( k == 0 ? -1.0 : 1.0 )*m.dt*g;
            const double w3[3] = { h*s[0], h*s[1], h*s[2] };
            double l3[3];
            d_tmulv( ( md & 2 ) ? RA : RB, w3, l3 );
            sd[0] += l3[0]; sd[1] += l3[1]; sd[2] += l3[2];
          }
          L.SV[3*slot] = sv[0]; L.SV[3*slot+1] = sv[1]; L.SV[3*slot+2] = sv[2];
          L.SD[3*slot] = sd[0]; L.SD[3*slot+1] = sd[1]; L.SD[3*slot+2] = sd[2];
        }
        const int ct = m.ci_type[RKFD_CI_CI( cinf )];
        is_rg = ct == RKFD_CONTACT_RIGID; is_el = ct == RKFD_CONTACT_ELASTIC;
      } else {
        L.act[j] = 0;
        L.asl[j] = 0;
      }
    }
    /* ordered compaction */
    const unsigned long long mrg = BALLOT( is_rg ), mel = BALLOT( is_el );
    const int prg = base_rg + __builtin_popcountll( mrg & below ), pel = base_el + __builtin_popcountll( mel & below );
    if( is_rg && prg < m.maxrg ) L.lrg[prg] = j;
    if( is_el && pel < m.maxact ) L.lel[pel] = j;
    if( __builtin_popcountll( mact ) + base_act > m.maxact ) ovf = 1;
    base_act += __builtin_popcountll( mact );
    if( base_act > m.maxact ) base_act = m.maxact;
    base_rg += __builtin_popcountll( mrg );
    base_el += __builtin_popcountll( mel );
  }
  if( lane == 0 ){
    if( base_rg > m.maxrg ){ base_rg = m.maxrg; ovf = 1; }   /* contact capacity exceeded */
    if( ovf ) L.cnt[CNT_OVF] = 1;
    L.cnt[CNT_NRG] = base_rg;
    L.cnt[CNT_NEL] = base_el < m.maxact ? base_el : m.maxact;
  }
  SYNC();
}

/* accumulate the contact forces CF of the listed contacts into the links' external
 * wrenches (rkFDContactForcePushWrench, reference src/rkfd_util.c:268-282): in world
 * coordinates the wrench on the owner link is (x x f, f), on the other link its negative.
 * lanes 0..5 own one component each and walk the list in order (deterministic). */
RKFD_DEV void rkfd_push_wrenches(const rkfdDevModel &m, const rkfdLds &L, const unsigned short *list, int n)
{
  const int lane = LANE();
  if( lane < 6 && n > 0 ){
    /* consecutive contacts usually act on the same two links (vertices of one shape pair): keep
     * the running sums in registers and touch LDS only when the link changes.  The next contact's
     * operands are fetched while the current one is summed. */
    int la = -1, lb = -1;
    double sa = 0, sb = 0;
    int jn = list[0], sln = L.asl[jn], cinfn = L.CIp[jn];
    double fn0 = L.CF[3*sln], fn1 = L.CF[3*sln+1], fn2 = L.CF[3*sln+2];
    double xn0 = L.CX[3*sln], xn1 = L.CX[3*sln+1], xn2 = L.CX[3*sln+2];
    for( int e=0; e<n; e++ ){
      const int cinf = cinfn;
      const double f[3] = { fn0, fn1, fn2 }, x[3] = { xn0, xn1, xn2 };
      if( e+1 < n ){
        jn = list[e+1]; sln = L.asl[jn]; cinfn = L.CIp[jn];
        fn0 = L.CF[3*sln]; fn1 = L.CF[3*sln+1]; fn2 = L.CF[3*sln+2];
        xn0 = L.CX[3*sln]; xn1 = L.CX[3*sln+1]; xn2 = L.CX[3*sln+2];
      }
      double w;
      if( lane < 3 ){
        double t[3]; d_cross( x, f, t );
        w = lane == 0 ? t[0] : ( lane == 1 ? t[1] : t[2] );
      } else {
        w = lane == 3 ? f[0] : ( lane == 4 ? f[1] : f[2] );
      }
      const int a = RKFD_CI_A( cinf ), bq = RKFD_CI_B( cinf );
      if( a != la ){ if( la >= 0 ) L.PB[6*la+lane] -= sa; la = a; sa = 0; }   /* bias force = -external force */
      if( bq != lb ){ if( lb >= 0 ) L.PB[6*lb+lane] += sb; lb = bq; sb = 0; }
      sa += w; sb += w;
    }
    if( la >= 0 ) L.PB[6*la+lane] -= sa;
    if( lb >= 0 ) L.PB[6*lb+lane] += sb;
  }
  SYNC();
}

/* rkFDContactForceModifyFriction (reference src/rkfd_util.c:239-266), one lane = one contact */
RKFD_DEV void d_modify_friction(const rkfdDevModel &m, const rkfdLds &L, int j, const double *vr, double *f, bool doUpRef)
{
  const int ci = RKFD_CI_CI( L.CIp[j] );
  double ax[9];
  d_load_axes( L, L.asl[j], ax );
  const double fn = d_dot( f, ax );
  const double f1 = d_dot( f, ax+3 ), f2 = d_dot( f, ax+6 );
  const double fs = sqrt( f1*f1 + f2*f2 );
  const double mu = L.typ[j] == RKFD_SF ? m.ci_sf[ci] : m.ci_kf[ci];
  if( !( fabs( fs ) < RKFD_DEV_TOL ) && fs > mu*fn ){
    const double vn = d_dot( vr, ax );
    double v[3] = { vr[0]-vn*ax[0], vr[1]-vn*ax[1], vr[2]-vn*ax[2] };
    const double vs = sqrt( d_dot( v, v ) );
    f[0] = fn*ax[0]; f[1] = fn*ax[1]; f[2] = fn*ax[2];
    if( !( fabs( vs ) < RKFD_DEV_TOL ) ){
      const double k = -( 1.0 - exp( -1.0*m.fric_w*vs ) )*m.ci_kf[ci]*fn/vs;
      f[0] += k*v[0]; f[1] += k*v[1]; f[2] += k*v[2];
    }
    if( doUpRef ){
      L.typ[j] = RKFD_KF;
      { const int ri = RIDX( j ), sl_ = L.asl[j]; L.REF[3*ri] = L.PRO[3*sl_]; L.REF[3*ri+1] = L.PRO[3*sl_+1]; L.REF[3*ri+2] = L.PRO[3*sl_+2]; }
    }
  } else {
    if( doUpRef ){
      L.typ[j] = RKFD_SF;
      if( m.has_slide ){ const int ri = RIDX( j ), sl_ = L.asl[j]; L.REF[3*ri] += L.SD[3*sl_]; L.REF[3*ri+1] += L.SD[3*sl_+1]; L.REF[3*ri+2] += L.SD[3*sl_+2]; }
    }
  }
}

/* rkFDSolverPenalty (reference src/rkfd_penalty.c:11-31), lane = elastic contact */
RKFD_DEV void rkfd_phase_penalty(const rkfdDevModel &m, const rkfdLds &L, bool doUpRef)
{
  const int lane = LANE();
  const int nel = L.cnt[CNT_NEL];
  if( lane < nel ){
    const int j = L.lel[lane], cinf = L.CIp[j], ci = RKFD_CI_CI( cinf );
    const double x[3] = { L.CX[3*L.asl[j]], L.CX[3*L.asl[j]+1], L.CX[3*L.asl[j]+2] };
    double va[3], vb[3], vr[3], f[3];
    d_point_vel( &L.V[6*RKFD_CI_A( cinf )], x, va );
    d_point_vel( &L.V[6*RKFD_CI_B( cinf )], x, vb );
    const double E = m.ci_e[ci], kv = -1.0*( m.ci_v[ci] + E*m.dt );
#pragma unroll
    for( int k=0; k<3; k++ ){
      vr[k] = va[k]-vb[k] + ( m.has_slide ? L.SV[3*L.asl[j]+k] : 0.0 );
      f[k] = -E*( x[k]-L.RW[3*L.asl[j]+k] ) + kv*vr[k];
    }
    if( d_dot( f, &L.AX[6*L.asl[j]] ) < 0.0 ){
      f[0] = f[1] = f[2] = 0;
    } else {
      d_modify_friction( m, L, j, vr, f, doUpRef );
    }
    { const int sl_ = L.asl[j]; L.CF[3*sl_] = f[0]; L.CF[3*sl_+1] = f[1]; L.CF[3*sl_+2] = f[2]; L.FS[sl_] = 1; }
  }
  SYNC();
  rkfd_push_wrenches( m, L, L.lel, nel );
}

/* velocity-dependent parts of the MLCP bias for rigid contact `lane` (lane = position in the rigid
 * list): bv[0..2] = axis . relative point velocity, bv[3..5] = axis . ( w x ( v_O + w x p ) of the
 * owner link minus that of the other link ).  Evaluated before the sweeps so that the link
 * velocities V need not outlive the contact phases (their LDS is reused for W = Ia c). */
RKFD_DEV void rkfd_phase_bvel(const rkfdDevModel &m, const rkfdLds &L, double *bv)
{
  const int lane = LANE();
  const int nc = L.cnt[CNT_NRG];
#pragma unroll
  for( int k=0; k<6; k++ ) bv[k] = 0;
  if( lane < nc ){
    const int j = L.lrg[lane], cinf = L.CIp[j], sl = L.asl[j];
    const int la = RKFD_CI_A( cinf ), lb = RKFD_CI_B( cinf );
    const double x[3] = { L.CX[3*sl], L.CX[3*sl+1], L.CX[3*sl+2] };
    double va[3], vb[3], ca[3], cb[3];
    d_point_vel( &L.V[6*la], x, va ); d_point_vel( &L.V[6*lb], x, vb );
    d_cross( &L.V[6*la], va, ca ); d_cross( &L.V[6*lb], vb, cb );
    double axl[9];
    d_load_axes( L, sl, axl );
#pragma unroll
    for( int i=0; i<3; i++ ){
      const double *ax = &axl[3*i];
      bv[i]   = ax[0]*( va[0]-vb[0] ) + ax[1]*( va[1]-vb[1] ) + ax[2]*( va[2]-vb[2] );
      if( m.has_slide ) bv[i] += ax[0]*L.SV[3*sl] + ax[1]*L.SV[3*sl+1] + ax[2]*L.SV[3*sl+2];
      bv[3+i] = ax[0]*( ca[0]-cb[0] ) + ax[1]*( ca[1]-cb[1] ) + ax[2]*( ca[2]-cb[2] );
    }
  }
  SYNC();
}

#endif /* RKFD_DEV_CONTACT_H */
