/* rkfd_dev_brf.h - part of the device code of the batched rkFDUpdate step (see rkfd_device.h):
 * BREAKABLE FLOAT JOINTS (RoKi's rk_joint_brfloat, un-vendored [UNVERIFIED-DEP]; reference example/model/wall.ztk:51-53,
 * example/chain/arm_wall_test.c; restated in oracle/rkfd_oracle.c: ejt / break_test).
 * Included by rkfd_device.h only, in this order; compiles for gfx950 and under the lane emulator.
 *
 * A link on such a joint has six coordinates like a float joint.  Until the joint breaks the link is rigidly attached to its
 * parent; at every COMMITTING evaluation the wrench the joint transmits is tested against the joint's thresholds, and once
 * either is passed the joint is a float joint from the next evaluation on.
 *
 * In the tables of the device model such a link IS a float joint (six coordinates, a Cholesky / frame slot, tree structure) and
 * additionally owns a pool slot.  What it is in one evaluation is decided per instance by L.BRK[link] (0 no such joint,
 * 1 unbroken, 2 broken) and told to the phases through the joint-type field of the link's packed info in LDS (L.LI):
 *   kinematics      : FLOAT either way (the link sits where its coordinates put it; its rates are zero while it is attached,
 *                     because its accelerations are)                                   - rkfd_brf_before_kinematics
 *   everything else : FIXED while unbroken                                            - rkfd_brf_after_kinematics
 *     sweep 2: a fixed joint hands its whole articulated inertia and bias force to the parent (pool slot); kept besides: the
 *              inertia (packed, in the link's Cholesky slot) and the bias (head of its frame slot) for the break test
 *     sweep 3: the parent's acceleration passes through, the six joint accelerations are zero
 *     probes : an unbroken joint is a level a probe path passes without leaving an innovation (S, U, 1/D zero); the table of
 *              where a path ends (TOP) is rebuilt from the joint types of the evaluation
 * The break test (rkfd_brf_break_test) needs the wrench the joint transmits to its link with the FINAL accelerations and all
 * external forces: W = Ia a + pA.  Whatever hangs on a breakable joint hangs on breakable or fixed joints (the host refuses
 * anything else), so the subtree of an unbroken joint is one rigid body as far as its unbroken joints reach: Ia and the bias pA of
 * the free motion come from sweep 2, a = a_free + delta a from the two forward sweeps, and the rigid contact forces solved in
 * between change pA by minus their wrenches on that body.
 */
#ifndef RKFD_DEV_BRF_H
#define RKFD_DEV_BRF_H

RKFD_DEV int rkfd_li_with_jt(int li, int jt){ return ( li & ~( 7 << 8 ) ) | ( jt << 8 ); }

/* the state of the launch: broken flags of this instance's joints -> L.BRK */
RKFD_DEV void rkfd_brf_load(const rkfdDevModel &m, const rkfdDevState &st, const rkfdLds &L, int b)
{
  const int lane = LANE();
  if( lane < m.nlink ){
    const int f = m.brf[lane];
    L.BRK[lane] = (unsigned char)( f ? ( st.brk[(size_t)b*m.nlink_model+m.orig[lane]] ? RKFD_BRF_BROKEN : RKFD_BRF_ATTACHED ) : RKFD_BRF_NONE );
  }
}
RKFD_DEV void rkfd_brf_store(const rkfdDevModel &m, const rkfdDevState &st, const rkfdLds &L, int b)
{
  const int lane = LANE();
  if( lane < m.nlink && L.BRK[lane] != RKFD_BRF_NONE ) st.brk[(size_t)b*m.nlink_model+m.orig[lane]] = L.BRK[lane] == RKFD_BRF_BROKEN;
}

RKFD_DEV void rkfd_brf_before_kinematics(const rkfdDevModel &m, const rkfdLds &L)
{
  const int lane = LANE();
  if( lane < m.nlink && L.BRK[lane] != RKFD_BRF_NONE ) L.LI[lane] = rkfd_li_with_jt( L.LI[lane], RKFD_JOINT_FLOAT );
  SYNC();
}
RKFD_DEV void rkfd_brf_after_kinematics(const rkfdDevModel &m, const rkfdLds &L)
{
  const int lane = LANE();
  const int NL = m.nlink;
  if( lane < NL && L.BRK[lane] == RKFD_BRF_ATTACHED ){
    L.LI[lane] = rkfd_li_with_jt( L.LI[lane], RKFD_JOINT_FIXED );
#pragma unroll
    for( int k=0; k<6; k++ ) L.S[6*lane+k] = 0.0;
  }
  SYNC();
  if( m.maxrg > 0 && lane < NL ){
    /* where a force on this link stops propagating upwards with the joint types of THIS evaluation: the nearest float joint at or
     * above it, else the root; 255 for a link that cannot move (fixed joints all the way up) */
    unsigned char *TOP = L.PL + NL*m.nlevel;
    int a = lane;
    while( RKFD_LI_JT( L.LI[a] ) != RKFD_JOINT_FLOAT && RKFD_LI_PAR( L.LI[a] ) >= 0 ) a = RKFD_LI_PAR( L.LI[a] );
    bool stat = RKFD_LI_JT( L.LI[a] ) == RKFD_JOINT_FIXED;
    for( int t=lane; stat && t!=a; t=RKFD_LI_PAR( L.LI[t] ) ) stat = RKFD_LI_JT( L.LI[t] ) == RKFD_JOINT_FIXED;
    TOP[lane] = (unsigned char)( stat ? 255 : a );
  }
  SYNC();
}
/* in front of the probes (contact phases): the quantities a probe reads at every level of its path are zero where the level is
 * an unbroken breakable joint (sweep 2 writes them for 1-DoF joints only) */
RKFD_DEV void rkfd_brf_before_probes(const rkfdDevModel &m, const rkfdLds &L)
{
  const int lane = LANE();
  if( lane < m.nlink && L.BRK[lane] == RKFD_BRF_ATTACHED ){
#pragma unroll
    for( int k=0; k<6; k++ ) L.U[6*lane+k] = 0.0;
    L.MS[3*lane+0] = 0.0; L.MS[3*lane+2] = 0.0;
  }
}

/* W += Ia x for the attached breakable joints, lane = link; Ia: the symmetric articulated inertia sweep 2 left packed in the link's
 * Cholesky slot, x = L.AC of the link, W: the head of the link's frame slot.  After the free forward sweep W holds the bias force
 * sweep 2 left there and x is the free acceleration; after the delta sweep x is the change of the acceleration the rigid contact
 * forces cause. */
RKFD_DEV void rkfd_brf_wrench_part(const rkfdDevModel &m, const rkfdLds &L)
{
  const int lane = LANE();
  const unsigned char *FSL = L.PL + m.nlink*m.nlevel + m.nlink;
  if( lane < m.nlink && L.BRK[lane] == RKFD_BRF_ATTACHED ){
    /* (the float-slot table lives with the path tables, which only worlds with a rigid contact capacity keep; otherwise count) */
    int fs = 0;
    if( m.maxrg > 0 ) fs = FSL[lane];
    else for( int l=0; l<lane; l++ ) fs += RKFD_LI_JT( m.linfo[l] ) == RKFD_JOINT_FLOAT;
    const double *Ia = &L.CHOL[21*fs];
    double *W = &L.XF[12*fs];
    double x[6], w[6];
#pragma unroll
    for( int k=0; k<6; k++ ){ x[k] = L.AC[6*lane+k]; w[k] = W[k]; }
#pragma unroll
    for( int r=0; r<6; r++ )
#pragma unroll
      for( int c=0; c<6; c++ ) w[r] = fma( Ia[r >= c ? RKFD_TRI( r, c ) : RKFD_TRI( c, r )], x[c], w[r] );
#pragma unroll
    for( int k=0; k<6; k++ ) W[k] = w[k];
  }
  SYNC();
}

/* the wrench w ((ang, lin) about the origin of the spatial coordinates) the joint of link `lane` transmits against its thresholds:
 * the force's norm, and the norm of the torque about the link origin p (kept by the kinematics in the link's frame slot) */
RKFD_DEV bool rkfd_brf_decide(const rkfdDevModel &m, const rkfdLds &L, int lane, const double *XF, const double *w)
{
  const double p[3] = { XF[9], XF[10], XF[11] };
  double pf[3];
  d_cross( p, w+3, pf );
  const double tq[3] = { w[0]-pf[0], w[1]-pf[1], w[2]-pf[2] };
  const double fn = sqrt( d_dot( w+3, w+3 ) ), tn = sqrt( d_dot( tq, tq ) );
  return fn > m.brk_f[lane] || tn > m.brk_t[lane];
}
/* float slot of a link (the table lives with the path tables, which only worlds with a rigid contact capacity keep; otherwise count) */
RKFD_DEV int rkfd_brf_fslot(const rkfdDevModel &m, const rkfdLds &L, int lane)
{
  if( m.maxrg > 0 ) return ( L.PL + m.nlink*m.nlevel + m.nlink )[lane];
  int fs = 0;
  for( int l=0; l<lane; l++ ) fs += RKFD_LI_JT( m.linfo[l] ) == RKFD_JOINT_FLOAT;
  return fs;
}

/* the break test, at the end of a committing evaluation.  rigid: the evaluation solved rigid contact forces (L.CF per slot of
 * the nc = L.cnt[CNT_NRG] rigid contact vertices L.lrg; elastic penalty wrenches were part of the bias all along) */
RKFD_DEV void rkfd_brf_break_test(const rkfdDevModel &m, const rkfdLds &L, bool rigid)
{
  const int lane = LANE();
  const unsigned char *FSL = L.PL + m.nlink*m.nlevel + m.nlink;
  /* every joint is tested against the states of THIS evaluation (which joints are attached decides which contact forces a
   * joint carries): the verdicts are written only after every lane has read what it needs */
  bool breaks = false;
  if( lane < m.nlink && L.BRK[lane] == RKFD_BRF_ATTACHED ){
    int fs = 0;
    if( m.maxrg > 0 ) fs = FSL[lane];
    else for( int l=0; l<lane; l++ ) fs += RKFD_LI_JT( m.linfo[l] ) == RKFD_JOINT_FLOAT;
    const double *XF = &L.XF[12*fs];
    double w[6];      /* (ang, lin) about the origin of the spatial coordinates */
#pragma unroll
    for( int k=0; k<6; k++ ) w[k] = XF[k];
    if( rigid ){
      const int nc = L.cnt[CNT_NRG];
      for( int c=0; c<nc; c++ ){
        const int j = L.lrg[c], cinf = L.CIp[j], sl = L.asl[j];
        const double x[3] = { L.CX[3*sl], L.CX[3*sl+1], L.CX[3*sl+2] }, f[3] = { L.CF[3*sl], L.CF[3*sl+1], L.CF[3*sl+2] };
        double t[3];
        d_cross( x, f, t );
#pragma unroll
        for( int sd=0; sd<2; sd++ ){
          /* is the side's link part of the rigid body this joint carries: up from it through attached breakable joints */
          int k = sd == 0 ? RKFD_CI_A( cinf ) : RKFD_CI_B( cinf );
          while( k != lane && k >= 0 && L.BRK[k] == RKFD_BRF_ATTACHED ) k = RKFD_LI_PAR( L.LI[k] );
          if( k == lane ){
            /* the force acts +f on the owner of the vertex (side 0), -f on the other link; the bias is minus the external wrench */
            const double sg = sd == 0 ? -1.0 : 1.0;
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

#endif /* RKFD_DEV_BRF_H */
