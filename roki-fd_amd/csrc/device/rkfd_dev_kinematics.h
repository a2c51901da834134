/* rkfd_dev_kinematics.h - part of the device code of the batched rkFDUpdate step (see rkfd_device.h):
 * phase: forward kinematics, link velocities, per-link inertia staging, bias forces, joint friction.
 * Included by rkfd_device.h only, in this order; compiles for gfx950 and under the lane emulator. */
#ifndef RKFD_DEV_KINEMATICS_H
#define RKFD_DEV_KINEMATICS_H

/* ------------------------------------------------------------------------ */
/* phase: forward kinematics, link velocities, per-link spatial inertia and bias terms.
 * Mirrors _rkFDConnectJointState (reference src/rkfd_sim.c:290-302) + the per-link set-up
 * of RoKi's ABA.  lane = link. */
template<bool prof> RKFD_DEV void rkfd_phase_kinematics(const rkfdDevModel &m, const rkfdLds &L, const rkfdLaneLink &ll, unsigned long long *pc)
{
  unsigned long long k0 = prof ? RKFD_CLOCK() : 0ull, k1;
#define KST(k) do{ if( prof ){ k1 = RKFD_CLOCK(); pc[k] += k1 - k0; k0 = k1; } }while(0)
  const int lane = LANE();
  const int NL = m.nlink;
  const bool on = lane < NL;
  const int i = on ? lane : 0;
  const int li = L.LI[i];
  const int jt = on ? RKFD_LI_JT( li ) : RKFD_JOINT_FIXED;
  const int off = RKFD_LI_OFF( li );
  double R[9], p[3], Rj[9], vJ[6], q1 = 0, qd1 = 0, qdf[6] = {0,0,0,0,0,0};
  int anc[RKFD_MAX_ROUND];
  {
    const int *ancp = RELOAD( m.anc );
#pragma unroll
    for( int r=0; r<RKFD_MAX_ROUND; r++ ) anc[r] = ( on && r < m.nround ) ? ancp[r*NL+i] : -1;
  }

  /* local (adjacent) transform = org frame * joint transform */
  {
    const double *Ro = &RELOAD( m.org )[12*i];
    double o[12];
#pragma unroll
    for( int k=0; k<12; k++ ) o[k] = Ro[k];
    Rj[0]=1; Rj[1]=0; Rj[2]=0; Rj[3]=0; Rj[4]=1; Rj[5]=0; Rj[6]=0; Rj[7]=0; Rj[8]=1;
#pragma unroll
    for( int k=0; k<9; k++ ) R[k] = o[k];
    p[0]=o[9]; p[1]=o[10]; p[2]=o[11];
    if( jt == RKFD_JOINT_REVOL ){
      double s, c;
      q1 = L.q[off];
      d_sincos( q1, &s, &c );
      double Rz[9] = { c,-s,0, s,c,0, 0,0,1 };
      d_mul33( o, Rz, R );
      qd1 = L.qd[off];
    } else if( jt == RKFD_JOINT_PRISM ){
      q1 = L.q[off];
      p[0] += q1*o[2]; p[1] += q1*o[5]; p[2] += q1*o[8];
      qd1 = L.qd[off];
    } else if( jt == RKFD_JOINT_FLOAT ){
      double qq[6], t[3];
#pragma unroll
      for( int k=0; k<6; k++ ){ qq[k] = L.q[off+k]; qdf[k] = L.qd[off+k]; }
      d_from_aa( qq+3, Rj );
      d_mul33( o, Rj, R );
      d_mulv( o, qq, t );
      p[0] += t[0]; p[1] += t[1]; p[2] += t[2];
    } else if( jt >= RKFD_DJT_SPHX ){
      /* spherical joint as three device links (RKFD_DJT_SPH*): the pseudo-links sit in the joint-origin frame, the real
       * link (SPHZ, last coordinate) is turned by the angle-axis vector of all three coordinates */
      qd1 = L.qd[off];
      if( jt == RKFD_DJT_SPHZ ){
        const double aa[3] = { L.q[off-2], L.q[off-1], L.q[off] };
        d_from_aa( aa, Rj );
        d_mul33( o, Rj, R );
      }
    }
  }
  if( on ){
#pragma unroll
    for( int k=0; k<6; k++ ){ L.XA[6*i+k] = R[k]; L.XB[6*i+k] = k < 3 ? R[6+k] : p[k-3]; }
  }
  SYNC();
  KST(16);
  /* pointer jumping: compose with the ancestor 2^r levels up */
#pragma unroll
  for( int r=0; r<RKFD_MAX_ROUND; r++ ){
    if( r >= m.nround ) break;
    const int a = anc[r];
    if( a >= 0 ){
      double Ra[9], pa[3], t[3];
#pragma unroll
      for( int k=0; k<6; k++ ) Ra[k] = L.XA[6*a+k];
#pragma unroll
      for( int k=0; k<3; k++ ){ Ra[6+k] = L.XB[6*a+k]; pa[k] = L.XB[6*a+3+k]; }
      d_mulv( Ra, p, t );
      p[0] = pa[0]+t[0]; p[1] = pa[1]+t[1]; p[2] = pa[2]+t[2];
      d_mul33( Ra, R, R );
    }
    SYNC();
    if( a >= 0 ){
#pragma unroll
      for( int k=0; k<6; k++ ){ L.XA[6*i+k] = R[k]; L.XB[6*i+k] = k < 3 ? R[6+k] : p[k-3]; }
    }
    SYNC();
  }
  /* Origin of the spatial coordinates.  All spatial quantities of an evaluation are Pluecker vectors about
   * ONE common point, which may be any point: inertias and bias forces about a far-away point lose
   * digits to the parallel-axis terms (error ~ distance^2), so the point is the anchor link's current
   * position - every link position, hence every contact point and lever arm derived from it below, is
   * taken relative to it.  Nothing spatial is carried from one evaluation to the next. */
  if( m.anchor >= 0 ){
    const double o0 = L.XB[6*m.anchor+3], o1 = L.XB[6*m.anchor+4], o2 = L.XB[6*m.anchor+5];
    SYNC();
    p[0] -= o0; p[1] -= o1; p[2] -= o2;
    if( on ){ L.XB[6*i+3] = p[0]; L.XB[6*i+4] = p[1]; L.XB[6*i+5] = p[2]; }
    SYNC();
  }
  KST(17);
  /* joint motion axis and joint velocity in world coordinates */
  double Row[9] = {1,0,0, 0,1,0, 0,0,1};   /* float joints: world orientation of the joint-origin frame */
  {
    double z[3] = { R[2], R[5], R[8] }, S[6] = {0,0,0,0,0,0};
#pragma unroll
    for( int k=0; k<6; k++ ) vJ[k] = 0;
    if( jt == RKFD_JOINT_REVOL ){
      S[0]=z[0]; S[1]=z[1]; S[2]=z[2]; d_cross( p, z, S+3 );
#pragma unroll
      for( int k=0; k<6; k++ ) vJ[k] = S[k]*qd1;
    } else if( jt == RKFD_JOINT_PRISM ){
      S[3]=z[0]; S[4]=z[1]; S[5]=z[2];
#pragma unroll
      for( int k=0; k<6; k++ ) vJ[k] = S[k]*qd1;
    } else if( jt >= RKFD_DJT_SPHX ){
      /* axis k of the joint-origin frame through the joint centre: that frame is the pseudo-links' own, and R Rj' for the real link */
      double ax[3] = { R[0], R[3], R[6] };
      if( jt == RKFD_DJT_SPHY ){ ax[0] = R[1]; ax[1] = R[4]; ax[2] = R[7]; }
      if( jt == RKFD_DJT_SPHZ ){
        /* third column of R Rj' = R ( third row of Rj )' */
        ax[0] = R[0]*Rj[6] + R[1]*Rj[7] + R[2]*Rj[8]; ax[1] = R[3]*Rj[6] + R[4]*Rj[7] + R[5]*Rj[8]; ax[2] = R[6]*Rj[6] + R[7]*Rj[7] + R[8]*Rj[8];
      }
      S[0]=ax[0]; S[1]=ax[1]; S[2]=ax[2]; d_cross( p, ax, S+3 );
#pragma unroll
      for( int k=0; k<6; k++ ) vJ[k] = S[k]*qd1;
    } else if( jt == RKFD_JOINT_FLOAT ){
      /* world orientation of the joint-origin frame: Row = R Rj' */
      double RjT[9] = { Rj[0],Rj[3],Rj[6], Rj[1],Rj[4],Rj[7], Rj[2],Rj[5],Rj[8] };
      double vw[3], ww[3], t[3];
      d_mul33( R, RjT, Row );
      d_mulv( Row, qdf, vw ); d_mulv( Row, qdf+3, ww );
      d_cross( p, ww, t );
      vJ[0]=ww[0]; vJ[1]=ww[1]; vJ[2]=ww[2];
      vJ[3]=vw[0]+t[0]; vJ[4]=vw[1]+t[1]; vJ[5]=vw[2]+t[2];
      /* for float joints S holds the world velocity of the joint-origin-frame rate (lin part),
       * needed later for the velocity-product term */
      S[0]=vw[0]; S[1]=vw[1]; S[2]=vw[2]; S[3]=ww[0]; S[4]=ww[1]; S[5]=ww[2];
    }
    if( on ){
#pragma unroll
      for( int k=0; k<6; k++ ){ L.S[6*i+k] = S[k]; L.V[6*i+k] = vJ[k]; }
    }
  }
  SYNC();
  KST(18);
  /* velocities: prefix sum of joint velocities along the path to the root */
  {
    double v[6];
#pragma unroll
    for( int k=0; k<6; k++ ) v[k] = vJ[k];
#pragma unroll
    for( int r=0; r<RKFD_MAX_ROUND; r++ ){
      if( r >= m.nround ) break;
      const int a = anc[r];
      if( a >= 0 ){
#pragma unroll
        for( int k=0; k<6; k++ ) v[k] += L.V[6*a+k];
      }
      SYNC();
      if( a >= 0 ){
#pragma unroll
        for( int k=0; k<6; k++ ) L.V[6*i+k] = v[k];
      }
      SYNC();
    }
    KST(19);
    /* velocity-product acceleration c = v x vJ (+ float-joint term) */
    double c[6];
    if( jt >= RKFD_DJT_SPHX ){
      /* spherical joint: its three axes are fixed in the parent-side frame, so the term of the whole joint is
       * v x ( S w ) with S w = the link's velocity minus that of the link in front of the joint; it sits on the real
       * link, the pseudo-links carry none */
      double vt[6] = {0,0,0,0,0,0};
      if( jt == RKFD_DJT_SPHZ ){
        const int p1 = RKFD_LI_PAR( li ), p2 = RKFD_LI_PAR( L.LI[p1] ), rp = RKFD_LI_PAR( L.LI[p2] );
#pragma unroll
        for( int k=0; k<6; k++ ) vt[k] = v[k] - ( rp >= 0 ? L.V[6*rp+k] : 0.0 );
      }
      d_crm( v, vt, c );
    } else
    d_crm( v, vJ, c );
    if( jt == RKFD_JOINT_FLOAT ){
      double vw[3] = { L.S[6*i], L.S[6*i+1], L.S[6*i+2] }, ww[3] = { vJ[0], vJ[1], vJ[2] }, t[3];
      d_cross( vw, ww, t );
      c[3] += t[0]; c[4] += t[1]; c[5] += t[2];
    }
    /* spatial inertia about the world origin and bias force */
    const double ms = RELOAD( m.mass )[i];
    double cw[3], Iw[9], t9[9], Ic[9], RT[9] = { R[0],R[3],R[6], R[1],R[4],R[7], R[2],R[5],R[8] };
    {
      const double *cm = &RELOAD( m.com )[3*i], *I0 = &RELOAD( m.inertia )[9*i];
      double cl[3] = { cm[0], cm[1], cm[2] };
#pragma unroll
      for( int k=0; k<9; k++ ) Ic[k] = I0[k];
      d_mulv( R, cl, cw );
      cw[0] += p[0]; cw[1] += p[1]; cw[2] += p[2];
      d_mul33( R, Ic, t9 ); d_mul33( t9, RT, Iw );
    }
    /* momentum h = I v about the world origin: h_lin = m ( v_O + w x r ), h_ang = Iw w + r x h_lin
     * (the 6x6 itself is rebuilt row by row inside sweep 2 from the staged Iw, r, m) */
    double h[6], pb[6];
    {
      double wxr[3], t3[3];
      d_cross( v, cw, wxr );
      h[3] = ms*( v[3]+wxr[0] ); h[4] = ms*( v[4]+wxr[1] ); h[5] = ms*( v[5]+wxr[2] );
      d_mulv( Iw, v, t3 );
      d_cross( cw, h+3, wxr );
      h[0] = t3[0]+wxr[0]; h[1] = t3[1]+wxr[1]; h[2] = t3[2]+wxr[2];
    }
    d_crf( v, h, pb );
    /* gravity as an explicit force at the centre of mass: f = (r x mg, mg) */
    {
      double g[3] = { 0, 0, -RKFD_G*ms }, ng[3];
      d_cross( cw, g, ng );
      pb[0] -= ng[0]; pb[1] -= ng[1]; pb[2] -= ng[2]; pb[5] -= g[2];
    }
    /* float joints: remember the world frame for sweep 3 (the X region is reused by the sweeps) */
    {
      const unsigned long long fm = BALLOT( on && jt == RKFD_JOINT_FLOAT );
      if( on && jt == RKFD_JOINT_FLOAT ){
        const int fs = __builtin_popcountll( fm & ( lane == 0 ? 0ull : ( ~0ull >> ( 64-lane ) ) ) );
#pragma unroll
        for( int k=0; k<9; k++ ) L.XF[12*fs+k] = Row[k];
        L.XF[12*fs+9] = p[0]; L.XF[12*fs+10] = p[1]; L.XF[12*fs+11] = p[2];
      }
    }
    if( on ){
      const double r2 = d_dot( cw, cw );
      L.IST[14*i+0] = Iw[0] + ms*( r2 - cw[0]*cw[0] ); L.IST[14*i+1] = Iw[1] - ms*cw[0]*cw[1]; L.IST[14*i+2] = Iw[2] - ms*cw[0]*cw[2];
      L.IST[14*i+3] = Iw[4] + ms*( r2 - cw[1]*cw[1] ); L.IST[14*i+4] = Iw[5] - ms*cw[1]*cw[2]; L.IST[14*i+5] = Iw[8] + ms*( r2 - cw[2]*cw[2] );
      L.IST[14*i+6] = ms*cw[0]; L.IST[14*i+7] = ms*cw[1]; L.IST[14*i+8] = ms*cw[2];
      L.IST[14*i+9] = -ms*cw[0]; L.IST[14*i+10] = -ms*cw[1]; L.IST[14*i+11] = -ms*cw[2];
      L.IST[14*i+12] = ms; L.IST[14*i+13] = 0.0;
#pragma unroll
      for( int k=0; k<6; k++ ){ L.C[6*i+k] = c[k]; L.PB[6*i+k] = pb[k]; }
    }
    /* joint friction and joint torque:
     * rkFDJointFriction / rkFDJointFrictionRevolDC (reference src/rkfd_util.c:318-387) */
    if( on ){
      double tau = 0, jm = 0;
      if( jt == RKFD_JOINT_REVOL || jt == RKFD_JOINT_PRISM ){
        const int mt = RKFD_LI_MT( li );
        double tin = 0, treg = 0, tf = 0;
        const double in = ll.min;
        if( mt == RKFD_MOTOR_DC ){
          const double gear = RELOAD( m.mot_gear )[i], admit = RELOAD( m.mot_admit )[i];
          const double gk = gear*RELOAD( m.mot_k )[i];
          jm = RELOAD( m.mot_inertia )[i];      /* reflected through the gear on the host */
          tin = admit*gk*d_clamp( in, RELOAD( m.mot_vmin )[i], RELOAD( m.mot_vmax )[i] );
          treg = admit*gk*gk*qd1;
          tf = jm*( -qd1/m.dt ) - tin + treg + ll.pivp;
          double fmax;
          if( ll.pivt == RKFD_SF ) fmax = RELOAD( m.sfric )[i];
          else {
            const double q = q1;      /* (L.q shares its storage with PB / C, written above) */
            const double sg = qd1 > 0 ? 1.0 : ( qd1 < 0 ? -1.0 : 0.0 );
            fmax = -RELOAD( m.stiff )[i]*q - RELOAD( m.visc )[i]*qd1 - RELOAD( m.coulomb )[i]*sg;
          }
          fmax = fabs( fmax );
          int newt;
          if( fabs( tf ) > fmax ){ tf = tf > 0 ? fmax : -fmax; newt = RKFD_KF; }
          else newt = RKFD_SF;
          /* the pivot type is committed by the caller when doUpRef (stored in MS slot 1 as a flag) */
          L.MS[3*i+1] = (double)newt;
        } else if( mt == RKFD_MOTOR_TRQ ){
          tin = d_clamp( in, RELOAD( m.mot_vmin )[i], RELOAD( m.mot_vmax )[i] );
        }
        tau = tin - treg + tf;
        /* driving torque without the inertia term + friction, for rkFDUpdateJointPrevDrivingTrq */
        L.MS[3*i+0] = tin - treg + tf;
      }
      L.MS[3*i+2] = tau;
    }
  }
  SYNC();
  KST(20);
#undef KST
}

#endif /* RKFD_DEV_KINEMATICS_H */
