/* rkfd_dev_step.h - part of the device code of the batched rkFDUpdate step (see rkfd_device.h):
 * one dynamics evaluation, the Runge-Kutta-Gill step and the per-instance driver (state load / store).
 * Included by rkfd_device.h only, in this order; compiles for gfx950 and under the lane emulator. */
#ifndef RKFD_DEV_STEP_H
#define RKFD_DEV_STEP_H

/* ------------------------------------------------------------------------ */
/* one dynamics evaluation: _rkFDUpdate / _rkFDUpdateRef (reference src/rkfd_sim.c:533-549).
 * Input L.q, L.qd; output L.acc (and contact / pivot state).  Returns nonzero when the model
 * needs a rigid solver that is not available on the device (wave-uniform). */
/* vqp: 0 = PGS only, 1 = also the Vert plugin's QP, 2 = the Volume plugin (kernel variants) */
template<bool prof, int vqp, bool pk> RKFD_DEV int rkfd_evaluate(const rkfdDevModel &m, const rkfdLds &L, rkfdLaneLink &ll, bool doUpRef, unsigned long long *pc)
{
  const int lane = LANE();
  int err = 0;
  unsigned long long t0 = 0, t1;
#define STAMP(k) do{ if( prof ){ t1 = RKFD_CLOCK(); pc[k] += t1 - t0; t0 = t1; } }while(0)
  if( prof ) t0 = RKFD_CLOCK();
  if( m.has_brf ) rkfd_brf_before_kinematics( m, L );
  rkfd_phase_kinematics<prof>( m, L, ll, pc );
  if( m.has_brf ) rkfd_brf_after_kinematics( m, L );
  STAMP(0);
  /* commit joint friction pivots (the reference does so inside rkFDJointFrictionRevolDC) */
  if( doUpRef && lane < m.nlink ){
    const int jt = RKFD_LI_JT( L.LI[lane] );
    if( ( jt == RKFD_JOINT_REVOL || jt == RKFD_JOINT_PRISM ) && RKFD_LI_MT( L.LI[lane] ) == RKFD_MOTOR_DC )
      ll.pivt = (int)L.MS[3*lane+1];
  }
  /* MS slot 0 carries (driving torque + friction) until sweep 2 overwrites it: keep a copy */
  double drv = 0;
  if( lane < m.nlink ) drv = L.MS[3*lane+0];
  SYNC();
  if( m.ncand > 0 ){
    rkfd_phase_collision( m, L );
    if( L.cnt[CNT_NEL] > 0 ) rkfd_phase_penalty( m, L, doUpRef );
  } else if( lane == 0 ){
    L.cnt[CNT_NRG] = 0; L.cnt[CNT_NEL] = 0;
  }
  SYNC();
  if( vqp == 2 ){
    if( m.vol_np > 0 ) rkfd_phase_volcol<prof>( m, L, pc );
    else if( lane == 0 ) L.cnt[CNT_NVP] = 0;
    SYNC();
  }
  double bv[6];
  rkfd_phase_bvel( m, L, bv );
  STAMP(1);
  /* rkChainUpdateABI (with rigid contacts this is rkFDUpdateAccBias) */
  rkfd_phase_sweep2<prof>( m, L, pc );
  STAMP(2);
  rkfd_phase_sweep3<false>( m, L );
  STAMP(3);
  bool solved = false;      /* rigid contact forces were solved in this evaluation */
  if( m.has_brf && doUpRef ) rkfd_brf_wrench_part( m, L );      /* bias + Ia x free acceleration */
  if( vqp == 2 ){
    if( L.cnt[CNT_NVP] > 0 ){
      SYNC();
      const double afree = lane < m.ndof ? L.acc[lane] : 0.0;
      rkfd_phase_volume<prof>( m, L, doUpRef, pc );
      STAMP(4);
      rkfd_phase_sweep3<true>( m, L );
      SYNC();
      solved = true;
      if( m.has_brf && doUpRef ) rkfd_brf_wrench_part( m, L );
      if( lane < m.ndof ) L.acc[lane] += afree;
      SYNC();
      STAMP(3);
    }
  } else if( L.cnt[CNT_NRG] > 0 ){
    if( m.solver == RKFD_SOLVER_MLCP || ( vqp == 1 && m.solver == RKFD_SOLVER_VERT && m.vert_rigid ) ){
      /* L.acc shares its LDS with the contact matrix: the free accelerations wait in a register (lane = dof) */
      SYNC();
      const double afree = lane < m.ndof ? L.acc[lane] : 0.0;
      /* contact forces, then their effect on the accelerations (rkChainUpdateCachedABI in the reference) */
      rkfd_phase_mlcp<prof, vqp == 1, pk>( m, L, bv, doUpRef, ll.dtask, pc );
      STAMP(4);
      rkfd_phase_sweep3<true>( m, L );
      SYNC();
      solved = true;
      if( m.has_brf && doUpRef ) rkfd_brf_wrench_part( m, L );      /* + Ia x the change of the acceleration */
      if( lane < m.ndof ) L.acc[lane] += afree;
      SYNC();
      STAMP(3);
    } else {
      err = 1;
    }
  }
  if( m.has_brf && doUpRef ){
    if( vqp == 2 ) rkfd_brf_break_test_vol( m, L, solved );
    else rkfd_brf_break_test( m, L, solved );
  }
  /* rkFDUpdateJointPrevDrivingTrq (reference src/rkfd_util.c:289-311), committing evaluation only */
  if( doUpRef && lane < m.nlink ){
    const int jt = RKFD_LI_JT( L.LI[lane] );
    if( jt == RKFD_JOINT_REVOL || jt == RKFD_JOINT_PRISM )
      ll.pivp = drv - m.mot_inertia[lane]*L.acc[RKFD_LI_OFF( L.LI[lane] )];
  }
  SYNC();
  STAMP(5);
#undef STAMP
  return err;
}

/* rkFDODECatDefault (reference src/rkfd_sim.c:306-320): q = q0 (+) k v.  lane = dof.
 * q0 and v are per-lane registers; the rotational part of float joints is composed by the
 * lane of the first angular dof (dofkind 1) through LDS. */
RKFD_DEV void rkfd_cat_dis(const rkfdDevModel &m, const rkfdLds &L, int dofkind, double q0, double k, double v)
{
  const int lane = LANE();
  const bool on = lane < m.ndof;
  const int kind = on ? dofkind : 0;
  if( on ){
    L.q[lane] = q0 + k*v;
    L.tmp[lane] = v;
    L.acc[lane] = q0;      /* acc is free at this point: used as scratch for q0 */
  }
  SYNC();
  if( on && kind == 1 ){
    double aa[3] = { k*L.tmp[lane], k*L.tmp[lane+1], k*L.tmp[lane+2] };
    double a0[3] = { L.acc[lane], L.acc[lane+1], L.acc[lane+2] };
    double Rk[9], R0[9], Rn[9], an[3];
    d_from_aa( aa, Rk ); d_from_aa( a0, R0 );
    d_mul33( Rk, R0, Rn );
    d_to_aa( Rn, an );
    L.q[lane] = an[0]; L.q[lane+1] = an[1]; L.q[lane+2] = an[2];
  }
  SYNC();
}

/* the whole step for one instance: load state, nsteps x rkFDUpdate (or a single evaluation),
 * store state.  mode 0: rkFDUpdate x nsteps; mode 1: rkFDUpdateInit (committing evaluation);
 * mode 2: evaluation without commit. */
/* The instance of this lane: with RKFD_W instances per wavefront, half h of workgroup wg simulates instance first + wg RKFD_W + h
 * and owns the h-th block of lds_instance bytes of the workgroup's LDS.  A half beyond the end of the batch part (odd count) keeps
 * in step by simulating the instance before it once more - the halves share every branch and barrier - and stores nothing. */
template<bool prof, int vqp, bool pk> RKFD_DEV void rkfd_instance(const rkfdDevModel &m_, const rkfdDevState &st, int b, void *ldsbase,
                            int mode, int nsteps, int *errflag, bool live = true, void *ldsshared = 0)
{
#ifdef RKFD_SPEC
  /* kernel compiled for ONE world (rkfdBatchSpecialize, hipRTC): its dimensions are literals, so the LDS layout,
   * loop bounds and table strides fold into immediates (config 4: 190 -> 84 SGPR spills, 54 -> 45 KB of code).  The length
   * of the sweep schedule stays a run-time value: as a literal it gets the sweeps unrolled, and 62 KB of code no longer
   * sit in the 64 KB instruction cache two CUs share (config 4 fell from 13 M to 3.9 M steps/s) */
  rkfdDevModel m = m_;
  m.nlink = RKFD_SPEC_NLINK; m.ndof = RKFD_SPEC_NDOF; m.ncand = RKFD_SPEC_NCAND; m.nlink_model = RKFD_SPEC_NLINK_MODEL;
  m.nlevel = RKFD_SPEC_NLEVEL; m.nround = RKFD_SPEC_NROUND; m.maxrg = RKFD_SPEC_MAXRG;
  m.npool = RKFD_SPEC_NPOOL; m.nfloat = RKFD_SPEC_NFLOAT; m.maxact = RKFD_SPEC_MAXACT; m.nside = RKFD_SPEC_NSIDE;
  m.npurow = RKFD_SPEC_NPUROW; m.pu_d0 = RKFD_SPEC_PU_D0; m.pu_alias = RKFD_SPEC_PU_ALIAS; m.vert_rigid = RKFD_SPEC_VERT_RIGID; m.qscr_alias = RKFD_SPEC_QSCR_ALIAS;
  m.has_slide = RKFD_SPEC_HAS_SLIDE; m.ma_size = RKFD_SPEC_MA_SIZE; m.ma_packed = RKFD_SPEC_MA_PACKED;
  m.max_iter = RKFD_SPEC_MAX_ITER; m.solver = RKFD_SPEC_SOLVER; m.pyramid = RKFD_SPEC_PYRAMID; m.anchor = RKFD_SPEC_ANCHOR;
  m.mlcp_mfma = RKFD_SPEC_MLCP_MFMA; m.has_brf = RKFD_SPEC_HAS_BRF; m.lds_shared = RKFD_SPEC_LDS_SHARED;
  m.vol_npair = RKFD_SPEC_VOL_NPAIR; m.vol_np = RKFD_SPEC_VOL_NP; m.vol_ncp = RKFD_SPEC_VOL_NCP; m.vol_pv = RKFD_SPEC_VOL_PV; m.vol_nf = RKFD_SPEC_VOL_NF;
#else
  const rkfdDevModel &m = m_;
#endif
  const int lane = LANE();
  const int ND = m.ndof, NL = m.nlink, NC = m.ncand;
  rkfdLds L;
  rkfd_lds_carve( &L, ldsbase, NL, ND, NC, 3*m.maxrg, m.nlevel, m.npool, m.nfloat, m.maxact, m.nside, m.pu_alias, m.npurow, m.vert_rigid, m.has_slide, m.ma_size,
                  vqp == 2 ? m.vol_np : 0, m.vol_ncp, m.vol_pv, m.vol_nf, m.pyramid, m.maxrg > 0, m.lds_shared > 0 ? ldsshared : 0 );
  /* the world's static tables, once per wavefront where the instances share them: the first instance of the wavefront fills them */
  const bool fills = !( m.lds_shared > 0 ) || HALF() == 0;
  if( m.lds_poison > 0 ){      /* (RKFD_DEBUG_POISON_LDS: see rkfd_devmodel.h) */
    for( int i=lane; i<m.lds_poison; i+=RKFD_WL ) ( (unsigned *)ldsbase )[i] = 0xffffffffu;
    SYNC();
  }
  if( lane == 0 ){
    L.cnt[CNT_OVF] = 0; L.cnt[CNT_QPF] = 0; if( vqp == 2 ) L.cnt[CNT_GRD] = 0;
    if( RKFD_GC_NEEDED( 3*m.maxrg ) ) L.GC[RKFD_GC_INTS-1] = -1;      /* no grouped layout remembered yet */
    if( NC > 0 ){ L.cnt[CNT_SRG] = 0; L.cnt[CNT_SEL] = 0; L.cnt[CNT_SN] = 0; }
  }

  /* load persistent state */
  double q = 0, qd = 0;
  int dofkind = 0;      /* 1: first angular coordinate of a float joint, 2: the other two */
  if( lane < ND ){ q = st.dis[(size_t)b*ND+lane]; qd = st.vel[(size_t)b*ND+lane]; dofkind = m.dofkind[lane]; }
  rkfdLaneLink ll; ll.min = 0; ll.pivp = 0; ll.pivt = 0; ll.dtask = -1;
  if( m.maxrg > 0 ){
    /* which link and component the lane's coordinate belongs to (once per launch: a scalar walk over the links) */
    const int li = lane < NL ? m.linfo[lane] : 0;
    for( int l=0; l<NL; l++ ){
      const int x = BCASTI( li, l ), jt = RKFD_LI_JT( x ), off = RKFD_LI_OFF( x );
      const int n = RKFD_JT_IS1( jt ) ? 1 : ( jt == RKFD_JOINT_FLOAT ? 6 : 0 );
      if( lane >= off && lane < off+n && lane < ND ) ll.dtask = l | ( ( lane-off ) << 8 );
    }
  }
  if( lane < NL ){
    if( fills ){
      L.LI[lane]   = m.linfo[lane];
      const int ch = m.child_idx[lane]; L.CHP[lane] = (unsigned short)( ch | ( ( m.pslot[ch]+1 ) << 8 ) );
    }
    const int lm = m.orig[lane];
    ll.min  = st.motor_in[(size_t)b*m.nlink_model+lm];
    ll.pivt = st.piv_type[(size_t)b*m.nlink_model+lm];
    ll.pivp = st.piv_prev[(size_t)b*m.nlink_model+lm];
  }
  if( m.maxrg > 0 && fills ){
    for( int k=lane; k<NL*( m.nlevel+3 ); k+=RKFD_WL ) L.PL[k] = (unsigned char)m.pathlink[k];
  }
  if( m.has_brf ) rkfd_brf_load( m, st, L, b );
  for( int c0=0, base=0; c0<NC; c0+=RKFD_WL ){
    const int j = c0 + lane;
    const bool onj = j < NC;
    int a = 0;
    if( onj ){
      if( fills ){ L.CIp[j] = m.cinfo[j]; L.CFO[j] = m.cand_foff[j]; }
      a = st.cv_active[(size_t)b*NC+j];
      L.typ[j] = a ? st.cv_type[(size_t)b*NC+j] : 0;
    }
    /* slots of the contacts alive at launch (candidate order; re-assigned by every collision pass) */
    const unsigned long long ma = BALLOT( a != 0 );
    const int sl = base + __builtin_popcountll( ma & ( lane == 0 ? 0ull : ( ~0ull >> ( 64-lane ) ) ) );
    base += __builtin_popcountll( ma );
    if( a && sl >= m.maxact ) a = 0;
    if( onj ){
      L.act[j] = a != 0;
      L.asl[j] = a ? sl : 0;
      if( a ){
#pragma unroll
        for( int k=0; k<3; k++ ) L.REF[3*sl+k] = st.cv_ref[((size_t)b*NC+j)*3+k];
      }
    }
  }
  SYNC();
  if( m.lds_shared > 0 ) SYNCW();
  int err = 0;
  /* phase-cycle counters exist only in the diagnostic instantiation (prof = true) */
  unsigned long long pc[prof ? RKFD_NPROF : 1];
#pragma unroll
  for( int k=0; k<( prof ? RKFD_NPROF : 1 ); k++ ) pc[k] = 0;
  const unsigned long long tstart = prof ? RKFD_CLOCK() : 0ull;
  {
    /* rkFDUpdate = zODE2Update (Runge-Kutta-Gill, 4 stage evaluations) + the committing
     * evaluation at the new state (reference src/rkfd_sim.c:560-566).  All five evaluations
     * run through ONE copy of rkfd_evaluate (stage loop) to keep the kernel inside the
     * instruction cache.  mode 1 / 2: a single evaluation at the current state. */
    /* Gill coefficients: (sqrt2-1)/2, (2-sqrt2)/2, -sqrt2/2, 1+sqrt2/2, 2-sqrt2, 2+sqrt2 */
    const double c21 = 0.20710678118654752440, c22 = 0.29289321881345247560, c31 = -0.70710678118654752440;
    const double c32 = 1.70710678118654752440, w2 = 0.58578643762690495120, w3 = 3.41421356237309504880;
    const bool on = lane < ND;
    const int nst = mode == 0 ? 5 : 1;
    const int ntot = mode == 0 ? nsteps*5 : 1;
    /* running sums instead of the four stage derivatives: F = weighted sum for the final update,
     * T = tangent of the next stage state, P = the part of the tangent after next known so far */
    double Fv = 0, Fa = 0, Tv = 0, Ta = 0, Pv = 0, Pa = 0;
    int stage = 0;
    for( int it=0; it<ntot; it++ ){
      double h = m.dt;
#ifndef RKFD_EMU
      asm volatile( "" : "+s"(h) );   /* keep h*coefficient products out of long-lived registers */
#endif
      const double k = stage == 1 ? 0.5*h : ( stage == 4 ? h/6.0 : h );
      const double xv = ( mode == 0 && stage > 0 ) ? fma( k, Ta, qd ) : qd;
      if( stage == 0 ){
        if( on ) L.q[lane] = q;
        SYNC();
      } else {
        rkfd_cat_dis( m, L, dofkind, q, k, Tv );
      }
      if( on ) L.qd[lane] = xv;
      if( stage == 4 ){ q = on ? L.q[lane] : 0.0; qd = xv; }
      SYNC();
      const bool doUp = mode == 0 ? stage == 4 : mode == 1;
      err |= rkfd_evaluate<prof, vqp, pk>( m, L, ll, doUp, pc );
      if( NC > 0 && mode == 0 && stage == 4 && lane == 0 ){     /* contact statistics of the step just committed */
        /* (the Volume plugin's rigid "contacts" are pairs in volumetric contact) */
        L.cnt[CNT_SRG] += vqp == 2 ? L.cnt[CNT_NVP] : L.cnt[CNT_NRG]; L.cnt[CNT_SEL] += L.cnt[CNT_NEL]; L.cnt[CNT_SN] += 1;
      }
      const double a = on ? L.acc[lane] : 0.0;
      if( stage == 0 ){ Fv = xv; Fa = a; Tv = xv; Ta = a; Pv = c21*xv; Pa = c21*a; }
      else if( stage == 1 ){ Fv = fma( w2, xv, Fv ); Fa = fma( w2, a, Fa ); Tv = fma( c22, xv, Pv ); Ta = fma( c22, a, Pa ); Pv = c31*xv; Pa = c31*a; }
      else if( stage == 2 ){ Fv = fma( w3, xv, Fv ); Fa = fma( w3, a, Fa ); Tv = fma( c32, xv, Pv ); Ta = fma( c32, a, Pa ); }
      else if( stage == 3 ){ Fv += xv; Fa += a; Tv = Fv; Ta = Fa; }
      SYNC();
      stage++; if( stage == nst ) stage = 0;
    }
  }
  if( prof && live && lane == 0 && st.prof ){
    pc[prof ? 7 : 0] = RKFD_CLOCK() - tstart;
#pragma unroll
    for( int k=0; k<( prof ? RKFD_NPROF : 1 ); k++ ) st.prof[(size_t)b*RKFD_NPROF+k] = pc[k];
  }
  /* store */
  if( live && lane < ND ){
    st.dis[(size_t)b*ND+lane] = q; st.vel[(size_t)b*ND+lane] = qd;
    st.acc[(size_t)b*ND+lane] = L.acc[lane];
  }
  /* (a spherical joint is three device links of one model link: the real one writes - the pseudo-links in front of it hold the same
   * untouched values, but three lanes storing to one address is a race all the same; found by ThreadSanitizer on the emulator) */
  if( live && lane < NL && RKFD_LI_JT( L.LI[lane] ) != RKFD_DJT_SPHX && RKFD_LI_JT( L.LI[lane] ) != RKFD_DJT_SPHY ){
    const int lm = m.orig[lane];
    st.piv_type[(size_t)b*m.nlink_model+lm] = ll.pivt;
    st.piv_prev[(size_t)b*m.nlink_model+lm] = ll.pivp;
  }
  if( m.has_brf && live ) rkfd_brf_store( m, st, L, b );
  /* contact state: the flag of every candidate, the rest only for those in contact (a candidate out of
   * contact has no state: type and anchor are re-initialised at its next first contact, and the
   * boundary reports zeros for it) */
  for( int j=lane; live && j<NC; j+=RKFD_WL ){
    const int a = L.act[j];
    st.cv_active[(size_t)b*NC+j] = a;
    if( a ){
      st.cv_type[(size_t)b*NC+j] = L.typ[j];
#pragma unroll
      for( int k=0; k<3; k++ ){
        st.cv_ref[((size_t)b*NC+j)*3+k] = L.REF[3*RIDX( j )+k];
        st.cv_f[((size_t)b*NC+j)*3+k] = L.FS[L.asl[j]] ? L.CF[3*L.asl[j]+k] : 0.0;
      }
    }
  }
  if( st.dbg && live ){
    /* debug dump: spatial accelerations (6/link) */
    if( lane < NL ){
      double *o = st.dbg + (size_t)b*st.dbg_stride;
      for( int k=0; k<6; k++ ) o[6*lane+k] = L.AC[6*lane+k];
    }
  }
  if( live && NC > 0 && mode == 0 && st.stat && lane < 3 )
    st.stat[(size_t)b*4+lane] += (unsigned int)L.cnt[CNT_SRG+lane];
  if( live && lane == 0 && errflag ){
    if( err ) *errflag = 1;              /* rigid contact with a solver that has no device path */
    if( L.cnt[CNT_OVF] ) *errflag = 2;   /* more rigid contacts than the configured capacity   */
    if( L.cnt[CNT_QPF] ) *errflag = 3;   /* the Vert QP ran out of iterations / basis history   */
    if( vqp == 2 && L.cnt[CNT_GRD] ) *errflag = 4;   /* Volume plugin: a pair that cannot be clipped came into contact */
  }
}

#endif /* RKFD_DEV_STEP_H */
