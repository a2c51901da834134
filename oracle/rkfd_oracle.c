/* rkfd_oracle.c - CPU restatement of the rkFDUpdate hot path (see rkfd_oracle.h:
 * TEST INFRASTRUCTURE ONLY, PARITY UNPINNED).
 *
 * Control flow follows the reference file by file:
 *   step driver ............ reference src/rkfd_sim.c:290-302,445-566
 *   contact bucketing ...... reference src/rkfd_cd.c:33-49
 *   relative vel/acc, friction cone clamp, wrench push, joint friction
 *                            reference src/rkfd_util.c (all)
 *   penalty force .......... reference src/rkfd_penalty.c:11-31
 *   MLCP / PGS ............. reference src/rkfd_mlcp.c (all)
 * Arithmetic the reference delegates to un-vendored RoKi / ZM / Zeo is restated
 * from the published algorithms, in RoKi's conventions (link-local frames,
 * (linear, angular) 6-D ordering, classical link accelerations, revolute /
 * prismatic axis = local z, float joint = position + angle-axis in the joint
 * origin frame).  Every such choice is tagged [UNVERIFIED-DEP] in DESIGN.md.
 */
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include "rkfd_oracle.h"

#define TOL RKFD_TOL
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#define is_tiny(x) ( fabs(x) < TOL )

/* ------------------------------------------------------------------------ */
/* small vector / matrix helpers (3-D, row-major 3x3) */
static void v3_zero(double *a){ a[0]=a[1]=a[2]=0; }
static void v3_copy(const double *a, double *b){ b[0]=a[0]; b[1]=a[1]; b[2]=a[2]; }
static void v3_add(const double *a, const double *b, double *c){ c[0]=a[0]+b[0]; c[1]=a[1]+b[1]; c[2]=a[2]+b[2]; }
static void v3_sub(const double *a, const double *b, double *c){ c[0]=a[0]-b[0]; c[1]=a[1]-b[1]; c[2]=a[2]-b[2]; }
static void v3_cat(double *a, double k, const double *b){ a[0]+=k*b[0]; a[1]+=k*b[1]; a[2]+=k*b[2]; }
static void v3_mul(const double *a, double k, double *c){ c[0]=k*a[0]; c[1]=k*a[1]; c[2]=k*a[2]; }
static double v3_dot(const double *a, const double *b){ return a[0]*b[0]+a[1]*b[1]+a[2]*b[2]; }
static double v3_norm(const double *a){ return sqrt( v3_dot(a,a) ); }
static void v3_cross(const double *a, const double *b, double *c)
{
  double x = a[1]*b[2]-a[2]*b[1], y = a[2]*b[0]-a[0]*b[2], z = a[0]*b[1]-a[1]*b[0];
  c[0]=x; c[1]=y; c[2]=z;
}
static void m3_mulv(const double *m, const double *v, double *r)
{
  double x = m[0]*v[0]+m[1]*v[1]+m[2]*v[2], y = m[3]*v[0]+m[4]*v[1]+m[5]*v[2], z = m[6]*v[0]+m[7]*v[1]+m[8]*v[2];
  r[0]=x; r[1]=y; r[2]=z;
}
static void m3_tmulv(const double *m, const double *v, double *r)
{
  double x = m[0]*v[0]+m[3]*v[1]+m[6]*v[2], y = m[1]*v[0]+m[4]*v[1]+m[7]*v[2], z = m[2]*v[0]+m[5]*v[1]+m[8]*v[2];
  r[0]=x; r[1]=y; r[2]=z;
}
static void m3_mul(const double *a, const double *b, double *c)
{
  double t[9]; int i, j;
  for( i=0; i<3; i++ ) for( j=0; j<3; j++ )
    t[3*i+j] = a[3*i]*b[j] + a[3*i+1]*b[3+j] + a[3*i+2]*b[6+j];
  memcpy( c, t, sizeof(t) );
}
static void m3_ident(double *m){ memset( m, 0, sizeof(double)*9 ); m[0]=m[4]=m[8]=1; }

/* rotation matrix from an angle-axis vector (Zeo zMat3DFromAA) */
static void m3_from_aa(const double *aa, double *m)
{
  double th = v3_norm( aa ), s, c, k, x, y, z;
  if( is_tiny( th ) ){ m3_ident( m ); return; }
  s = sin(th); c = cos(th); k = 1-c;
  x = aa[0]/th; y = aa[1]/th; z = aa[2]/th;
  m[0] = c+k*x*x;   m[1] = k*x*y-s*z; m[2] = k*x*z+s*y;
  m[3] = k*x*y+s*z; m[4] = c+k*y*y;   m[5] = k*y*z-s*x;
  m[6] = k*x*z-s*y; m[7] = k*y*z+s*x; m[8] = c+k*z*z;
}
/* angle-axis vector of a rotation matrix (Zeo zMat3DToAA) */
static void m3_to_aa(const double *m, double *aa)
{
  double l[3], a, th;
  l[0] = m[7]-m[5]; l[1] = m[2]-m[6]; l[2] = m[3]-m[1];
  a = v3_norm( l );
  th = atan2( a, m[0]+m[4]+m[8]-1.0 );
  if( is_tiny( a ) ){ v3_zero( aa ); return; }
  v3_mul( l, th/a, aa );
}

/* orthonormal complement of a unit normal: tangent 1 from the coordinate axis with the
 * smallest |component| (first on ties), tangent 2 = n x t1.  [UNVERIFIED-DEP] */
static void ortho_space(const double *n, double *t1, double *t2)
{
  int k = 0; double e[3] = {0,0,0}, d, l;
  if( fabs(n[1]) < fabs(n[k]) ) k = 1;
  if( fabs(n[2]) < fabs(n[k]) ) k = 2;
  e[k] = 1.0;
  d = v3_dot( e, n );
  t1[0] = e[0]-d*n[0]; t1[1] = e[1]-d*n[1]; t1[2] = e[2]-d*n[2];
  l = v3_norm( t1 );
  t1[0] /= l; t1[1] /= l; t1[2] /= l;
  v3_cross( n, t1, t2 );
}

/* ------------------------------------------------------------------------ */
typedef struct {
  double Ra[9], pa[3];   /* adjacent frame: link w.r.t. parent */
  double Rj[9];          /* float joint: rotation of the joint displacement */
  double R[9], p[3];     /* world frame */
  double v[6];           /* velocity, link frame (lin, ang) */
  double a[6];           /* classical acceleration, link frame (lin, ang) */
  double gam[6];         /* velocity-product acceleration */
  double M6[36];         /* spatial inertia about link origin, link frame */
  double IA[36];         /* articulated inertia */
  double U[6], D;        /* 1-DoF joints: IA S, S'IA S + motor inertia */
  double L6[36];         /* float joints: Cholesky factor of IA */
  double tau, tf, jm;    /* joint torque, friction torque, motor inertia (1-DoF) */
  double qd, q;          /* 1-DoF joint rate / displacement */
  double qdf[6];         /* float joint rate; spherical joint: angular rate in [3..5] */
  double S3[18], U3[18], Di3[9];   /* spherical joints: motion subspace (6x3, link frame), IA S, (S'IA S)^-1 */
} Link;

/* Volume plugin (rkfd_oracle_volume.h): a colliding rigid pair with its intersection volume and contact-plane conditions */
#define VOL_MAXPV 48      /* vertices of one clipped face polygon */
#define VOL_MAXCP 96      /* contact-plane conditions of one pair */

typedef struct { double v[3], n[3], r[2], s[2], th; } VolCP;
struct VolPair_ {
  int pair, ci, la, lb, sa, sb;
  double norm[3], axis[9], center[3], volume;
  int ntri, captri; double *tri;      /* colvol: 12 doubles per triangle (3 vertices, face normal) */
  int ncp; VolCP cp[VOL_MAXCP];
  double wrench[6];
  double q[36], c[6];                 /* the pair's 6x6 objective and linear term (kept for the tests) */
};
typedef struct VolPair_ VolPair;

struct rkfdOracle {
  const rkfdModel *m;
  int nl, n, ncand;
  double t;
  double *dis, *vel, *acc, *motor_in;
  int *piv_type; double *piv_prev;
  int *broken;           /* [nl] breakable float joints: 1 once the joint has broken (state, like the friction pivots) */
  int *cv_active, *cv_type; double *cv_ref, *cv_f;
  Link *lk;
  /* bias arrays (SoA so that save / restore are memcpy) */
  double *beta0, *ext, *pA, *u, *contrib, *csum;      /* u: 6 per link (1-DoF uses [0]) */
  double *s_beta0, *s_pA, *s_u, *s_contrib, *s_csum;  /* saved by "SaveABIAccBias" */
  /* per-candidate evaluation data */
  double *cx, *crefw, *cnorm, *caxis, *cpro, *cvel;
  int *clinkA, *clinkB, *cci;
  int nel, nrg; int *el, *rg;
  /* MLCP workspace */
  int mcap; double *ma, *mb, *mt, *mf;
  int last_nc;
  int last_qp_iter;      /* KKT solves of the last Vert QP (diagnostic) */
  int qp_cycle_stops;    /* how many Vert QPs so far were ended by the circulation check (diagnostic) */
  unsigned char *vol_raw;  /* Volume plugin: [nshape] 1 = the shape is not convex - its rigid pairs are GUARDED (see rkfdOracleCreate) */
  int vol_guard_hits;    /* ... evaluations x pairs in which a guarded pair was found in collision (and left without a force) */
  int qp_n, qp_mc;       /* the last Vert QP, kept for the tests: sizes, then q (n*n), c (n), nf (mc*n), ans (n), idx (mc) */
  double *qp_q, *qp_c, *qp_nf, *qp_ans; int *qp_idx;
  /* RKG workspace */
  double *k_v[4], *k_a[4], *xd, *xv, *tv, *ta;
  /* Volume plugin (rkfd_oracle_volume.h): face loops of the shapes, the colliding rigid pairs of this evaluation, the
   * stick / slip type per model pair, LPs that failed twice (diagnostic) */
  int vol_ready, *fl_off, *fl_idx, nvp, *vp_type, vol_lp_fail;
  VolPair *vp;
};

/* ------------------------------------------------------------------------ */
static void *zalloc(size_t n){ return calloc( n ? n : 1, 1 ); }

rkfdOracle *rkfdOracleCreate(const rkfdModel *m)
{
  rkfdOracle *o;
  int i, j, nl = m->nlink, n = m->ndof, nc = m->ncand, k;

  o = (rkfdOracle *)zalloc( sizeof(rkfdOracle) );
  /* The Volume plugin's intersection volumes are formed by clipping CONVEX shapes [UNVERIFIED-DEP, rkfd_oracle_volume.h].  A rigid
   * pair with a shape that is not convex (the body meshes of the reference's mighty.ztk) is GUARDED: the plugin's own collision
   * test - a vertex of one shape behind every face plane of the other - runs for it, a hit is counted (rkfdOracleVolumeGuardHits)
   * and the pair is left without a contact force; the device reports the same condition as status 4. */
  o->vol_raw = (unsigned char *)zalloc( m->nshape > 0 ? m->nshape : 1 );
  if( m->solver == RKFD_SOLVER_VOLUME ){
    int sh, f, v;
    for( sh=0; sh<m->nshape; sh++ )
      for( f=m->shape_foff[sh]; f<m->shape_foff[sh+1] && !o->vol_raw[sh]; f++ )
        for( v=m->shape_voff[sh]; v<m->shape_voff[sh+1]; v++ )
          if( v3_dot( &m->planes[4*f], &m->verts[3*v] ) - m->planes[4*f+3] > 1e-9 ){ o->vol_raw[sh] = 1; break; }
  }

  o->m = m; o->nl = nl; o->n = n; o->ncand = nc;
  o->dis = zalloc( sizeof(double)*n ); o->vel = zalloc( sizeof(double)*n ); o->acc = zalloc( sizeof(double)*n );
  o->motor_in = zalloc( sizeof(double)*nl );
  o->piv_type = zalloc( sizeof(int)*nl ); o->piv_prev = zalloc( sizeof(double)*nl );
  o->broken = zalloc( sizeof(int)*nl );
  o->cv_active = zalloc( sizeof(int)*nc ); o->cv_type = zalloc( sizeof(int)*nc );
  o->cv_ref = zalloc( sizeof(double)*3*nc ); o->cv_f = zalloc( sizeof(double)*3*nc );
  o->lk = zalloc( sizeof(Link)*nl );
  o->beta0 = zalloc( sizeof(double)*6*nl ); o->ext = zalloc( sizeof(double)*6*nl );
  o->pA = zalloc( sizeof(double)*6*nl ); o->u = zalloc( sizeof(double)*6*nl );
  o->contrib = zalloc( sizeof(double)*6*nl ); o->csum = zalloc( sizeof(double)*6*nl );
  o->s_beta0 = zalloc( sizeof(double)*6*nl ); o->s_pA = zalloc( sizeof(double)*6*nl ); o->s_u = zalloc( sizeof(double)*6*nl );
  o->s_contrib = zalloc( sizeof(double)*6*nl ); o->s_csum = zalloc( sizeof(double)*6*nl );
  o->cx = zalloc( sizeof(double)*3*nc ); o->crefw = zalloc( sizeof(double)*3*nc ); o->cnorm = zalloc( sizeof(double)*3*nc );
  o->caxis = zalloc( sizeof(double)*9*nc ); o->cpro = zalloc( sizeof(double)*3*nc ); o->cvel = zalloc( sizeof(double)*3*nc );
  o->clinkA = zalloc( sizeof(int)*nc ); o->clinkB = zalloc( sizeof(int)*nc ); o->cci = zalloc( sizeof(int)*nc );
  o->el = zalloc( sizeof(int)*nc ); o->rg = zalloc( sizeof(int)*nc );
  o->mcap = nc;
  o->ma = zalloc( sizeof(double)*9*nc*nc ); o->mb = zalloc( sizeof(double)*3*nc );
  o->mt = zalloc( sizeof(double)*3*nc ); o->mf = zalloc( sizeof(double)*3*nc );
  for( k=0; k<4; k++ ){ o->k_v[k] = zalloc( sizeof(double)*n ); o->k_a[k] = zalloc( sizeof(double)*n ); }
  o->xd = zalloc( sizeof(double)*n ); o->xv = zalloc( sizeof(double)*n );
  o->tv = zalloc( sizeof(double)*n ); o->ta = zalloc( sizeof(double)*n );

  /* constant spatial inertia about the link origin, (lin, ang) ordering:
   *   [ m 1      -m [c]x ]
   *   [ m [c]x    Ic - m [c]x [c]x ]                                            */
  for( i=0; i<nl; i++ ){
    double ms = m->mass[i]; const double *c = &m->com[3*i], *Ic = &m->inertia[9*i];
    double cx[9] = { 0,-c[2],c[1], c[2],0,-c[0], -c[1],c[0],0 }, cc[9];
    double *M = o->lk[i].M6;
    m3_mul( cx, cx, cc );
    for( j=0; j<3; j++ ) for( k=0; k<3; k++ ){
      M[6*j+k]       = ( j==k ? ms : 0.0 );
      M[6*j+3+k]     = -ms*cx[3*j+k];
      M[6*(3+j)+k]   =  ms*cx[3*j+k];
      M[6*(3+j)+3+k] = Ic[3*j+k] - ms*cc[3*j+k];
    }
    /* friction pivots start sticking (reference src/rkfd_sim.c:157-175) */
    o->piv_type[i] = RKFD_SF; o->piv_prev[i] = 0;
  }
  return o;
}

void rkfdOracleDestroy(rkfdOracle *o)
{
  int k;
  if( !o ) return;
  free( o->qp_q ); free( o->qp_c ); free( o->qp_nf ); free( o->qp_ans ); free( o->qp_idx );
  free( o->dis ); free( o->vel ); free( o->acc ); free( o->motor_in ); free( o->vol_raw ); free( o->piv_type ); free( o->piv_prev ); free( o->broken );
  free( o->cv_active ); free( o->cv_type ); free( o->cv_ref ); free( o->cv_f ); free( o->lk );
  free( o->beta0 ); free( o->ext ); free( o->pA ); free( o->u ); free( o->contrib ); free( o->csum );
  free( o->s_beta0 ); free( o->s_pA ); free( o->s_u ); free( o->s_contrib ); free( o->s_csum );
  free( o->cx ); free( o->crefw ); free( o->cnorm ); free( o->caxis ); free( o->cpro ); free( o->cvel );
  free( o->clinkA ); free( o->clinkB ); free( o->cci ); free( o->el ); free( o->rg );
  free( o->ma ); free( o->mb ); free( o->mt ); free( o->mf );
  for( k=0; k<4; k++ ){ free( o->k_v[k] ); free( o->k_a[k] ); }
  free( o->xd ); free( o->xv ); free( o->tv ); free( o->ta );
  if( o->vol_ready ){
    for( k=0; k<o->m->npair; k++ ) free( o->vp[k].tri );
    free( o->vp ); free( o->vp_type ); free( o->fl_off ); free( o->fl_idx );
  }
  free( o );
}

void rkfdOracleSetState(rkfdOracle *o, const double *dis, const double *vel)
{
  memcpy( o->dis, dis, sizeof(double)*o->n );
  memcpy( o->vel, vel, sizeof(double)*o->n );
}
void rkfdOracleGetState(const rkfdOracle *o, double *dis, double *vel, double *acc)
{
  if( dis ) memcpy( dis, o->dis, sizeof(double)*o->n );
  if( vel ) memcpy( vel, o->vel, sizeof(double)*o->n );
  if( acc ) memcpy( acc, o->acc, sizeof(double)*o->n );
}
void rkfdOracleSetMotorInput(rkfdOracle *o, const double *input){ memcpy( o->motor_in, input, sizeof(double)*o->nl ); }
double rkfdOracleTime(const rkfdOracle *o){ return o->t; }

void rkfdOracleGetContact(const rkfdOracle *o, int *active, int *type, double *ref, double *f)
{
  if( active ) memcpy( active, o->cv_active, sizeof(int)*o->ncand );
  if( type ) memcpy( type, o->cv_type, sizeof(int)*o->ncand );
  if( ref ) memcpy( ref, o->cv_ref, sizeof(double)*3*o->ncand );
  if( f ) memcpy( f, o->cv_f, sizeof(double)*3*o->ncand );
}
void rkfdOracleSetContact(rkfdOracle *o, const int *active, const int *type, const double *ref)
{
  memcpy( o->cv_active, active, sizeof(int)*o->ncand );
  memcpy( o->cv_type, type, sizeof(int)*o->ncand );
  memcpy( o->cv_ref, ref, sizeof(double)*3*o->ncand );
}
void rkfdOracleGetPivot(const rkfdOracle *o, int *type, double *prev_trq)
{
  if( type ) memcpy( type, o->piv_type, sizeof(int)*o->nl );
  if( prev_trq ) memcpy( prev_trq, o->piv_prev, sizeof(double)*o->nl );
}
void rkfdOracleSetPivot(rkfdOracle *o, const int *type, const double *prev_trq)
{
  memcpy( o->piv_type, type, sizeof(int)*o->nl );
  memcpy( o->piv_prev, prev_trq, sizeof(double)*o->nl );
}
void rkfdOracleGetBroken(const rkfdOracle *o, int *broken){ memcpy( broken, o->broken, sizeof(int)*o->nl ); }
void rkfdOracleSetBroken(rkfdOracle *o, const int *broken){ int i; for( i=0; i<o->nl; i++ ) o->broken[i] = broken[i] != 0 && o->m->jtype[i] == RKFD_JOINT_BRFLOAT; }

/* BREAKABLE FLOAT JOINT [RoKi rk_joint_brfloat, un-vendored; UNVERIFIED-DEP - restated from the model files and drivers that use it:
 * reference example/model/wall.ztk:51-53 (jointtype: breakablefloat, forcethreshold, torquethreshold), example/chain/arm_wall_test.c].
 * Six coordinates like a float joint.  Unbroken, the link is rigidly attached to its parent: in the ABA it is a fixed joint (its
 * whole articulated inertia and bias pass to the parent, its joint acceleration is zero, so its rates and coordinates stay as
 * set); broken, it is a float joint.  The joint type the dynamics see: */
static int ejt(const rkfdOracle *o, int i)
{
  const int jt = o->m->jtype[i];
  if( jt != RKFD_JOINT_BRFLOAT ) return jt;
  return o->broken[i] ? RKFD_JOINT_FLOAT : RKFD_JOINT_FIXED;
}
/* the kinematics see a float joint either way (a rigidly attached link sits where its six coordinates put it) */
static int kjt(const rkfdOracle *o, int i){ return o->m->jtype[i] == RKFD_JOINT_BRFLOAT ? RKFD_JOINT_FLOAT : o->m->jtype[i]; }

void rkfdOracleGetLinkFrames(const rkfdOracle *o, double *R, double *p)
{
  int i;
  for( i=0; i<o->nl; i++ ){ memcpy( R+9*i, o->lk[i].R, sizeof(double)*9 ); memcpy( p+3*i, o->lk[i].p, sizeof(double)*3 ); }
}
void rkfdOracleGetLinkVelAcc(const rkfdOracle *o, double *vel, double *acc)
{
  int i;
  for( i=0; i<o->nl; i++ ){
    if( vel ) memcpy( vel+6*i, o->lk[i].v, sizeof(double)*6 );
    if( acc ) memcpy( acc+6*i, o->lk[i].a, sizeof(double)*6 );
  }
}
int rkfdOracleGetMLCP(const rkfdOracle *o, double *a, double *b, double *f, int cap)
{
  int n3 = 3*o->last_nc;
  if( n3 > cap ) return -1;
  if( a ) memcpy( a, o->ma, sizeof(double)*n3*n3 );
  if( b ) memcpy( b, o->mb, sizeof(double)*n3 );
  if( f ) memcpy( f, o->mf, sizeof(double)*n3 );
  return o->last_nc;
}

/* ------------------------------------------------------------------------ */
/* _rkFDConnectJointState (reference src/rkfd_sim.c:290-302):
 * rkChainFK + rkChainSetJointVelAll + rkChainUpdateVel [RoKi, restated] */
static void connect_state(rkfdOracle *o, const double *dis, const double *vel)
{
  const rkfdModel *m = o->m;
  int i;
  for( i=0; i<o->nl; i++ ){
    Link *l = &o->lk[i];
    const double *Ro = &m->org[12*i], *po = Ro+9;
    const double *q = dis + m->dofoff[i], *qd = vel + m->dofoff[i];
    int par = m->parent[i];
    double wp[3], vp[3], t[3];

    /* adjacent frame = org frame * joint transform */
    switch( kjt( o, i ) ){
    case RKFD_JOINT_REVOL: {
      double s = sin(q[0]), c = cos(q[0]);
      double Rz[9] = { c,-s,0, s,c,0, 0,0,1 };
      m3_mul( Ro, Rz, l->Ra ); v3_copy( po, l->pa );
      l->q = q[0]; l->qd = qd[0];
    } break;
    case RKFD_JOINT_PRISM: {
      double z[3] = { Ro[2], Ro[5], Ro[8] };
      memcpy( l->Ra, Ro, sizeof(double)*9 );
      v3_copy( po, l->pa ); v3_cat( l->pa, q[0], z );
      l->q = q[0]; l->qd = qd[0];
    } break;
    case RKFD_JOINT_FLOAT: {
      m3_from_aa( q+3, l->Rj );
      m3_mul( Ro, l->Rj, l->Ra );
      m3_mulv( Ro, q, t ); v3_add( po, t, l->pa );
      memcpy( l->qdf, qd, sizeof(double)*6 );
    } break;
    case RKFD_JOINT_SPHER: {
      /* the rotational half of a float joint: displacement = angle-axis vector, rate = angular velocity, both in the
       * joint-origin (org) frame  [UNVERIFIED-DEP, same convention as the float joint] */
      m3_from_aa( q, l->Rj );
      m3_mul( Ro, l->Rj, l->Ra ); v3_copy( po, l->pa );
      memset( l->qdf, 0, sizeof(double)*3 ); memcpy( l->qdf+3, qd, sizeof(double)*3 );
    } break;
    default:
      memcpy( l->Ra, Ro, sizeof(double)*9 ); v3_copy( po, l->pa );
    }
    /* world frame and link-frame velocity from the parent */
    if( par < 0 ){
      memcpy( l->R, l->Ra, sizeof(double)*9 ); v3_copy( l->pa, l->p );
      memset( l->v, 0, sizeof(double)*6 );
      v3_zero( wp );
    } else {
      Link *pl = &o->lk[par];
      m3_mul( pl->R, l->Ra, l->R );
      m3_mulv( pl->R, l->pa, t ); v3_add( pl->p, t, l->p );
      /* v = Ra'( v_p + w_p x pa ), w = Ra' w_p */
      v3_cross( pl->v+3, l->pa, t ); v3_add( pl->v, t, vp );
      m3_tmulv( l->Ra, vp, l->v );
      m3_tmulv( l->Ra, pl->v+3, l->v+3 );
      v3_copy( pl->v+3, wp );
    }
    /* velocity-product acceleration (classical accelerations):
     *   lin: Ra'( w_p x ( w_p x pa ) ) (+ joint Coriolis), ang: w' x (joint angular rate) */
    {
      double wpa[3], cen[3], wl[3];
      v3_cross( wp, l->pa, wpa ); v3_cross( wp, wpa, cen );
      m3_tmulv( l->Ra, cen, l->gam );
      v3_zero( l->gam+3 );
      v3_copy( l->v+3, wl ); /* w' = Ra' w_p, before the joint's own rate is added */
      switch( kjt( o, i ) ){
      case RKFD_JOINT_REVOL: {
        double zq[3] = { 0, 0, l->qd };
        v3_cross( wl, zq, l->gam+3 );
        l->v[5] += l->qd;
      } break;
      case RKFD_JOINT_PRISM: {
        double zq[3] = { 0, 0, l->qd }, c2[3];
        v3_cross( wl, zq, c2 ); v3_cat( l->gam, 2.0, c2 );
        l->v[2] += l->qd;
      } break;
      case RKFD_JOINT_FLOAT: {
        /* joint rate (v_j, w_j) is expressed in the org frame; w_p in org frame = Ro' w_p */
        double wo[3], c2[3], c3[3], vj[3], wj[3];
        m3_tmulv( Ro, wp, wo );
        v3_cross( wo, l->qdf, c2 );   m3_tmulv( l->Rj, c2, c3 ); v3_cat( l->gam, 2.0, c3 );
        v3_cross( wo, l->qdf+3, c2 ); m3_tmulv( l->Rj, c2, l->gam+3 );
        m3_tmulv( l->Rj, l->qdf, vj );   v3_add( l->v, vj, l->v );
        m3_tmulv( l->Rj, l->qdf+3, wj ); v3_add( l->v+3, wj, l->v+3 );
      } break;
      case RKFD_JOINT_SPHER: {
        double wo[3], c2[3], wj[3];
        int k, a_;
        m3_tmulv( Ro, wp, wo );
        v3_cross( wo, l->qdf+3, c2 ); m3_tmulv( l->Rj, c2, l->gam+3 );
        m3_tmulv( l->Rj, l->qdf+3, wj ); v3_add( l->v+3, wj, l->v+3 );
        /* motion subspace in the link frame: the org-frame axes seen from the link, angular rows only */
        memset( l->S3, 0, sizeof(l->S3) );
        for( k=0; k<3; k++ ) for( a_=0; a_<3; a_++ ) l->S3[3*( 3+k )+a_] = l->Rj[3*a_+k];   /* S3[row 3+k][col a] = (Rj')[k][a] */
      } break;
      default: break;
      }
    }
  }
}

/* ------------------------------------------------------------------------ */
/* motor model [RoKi rk_motor_dc / rk_motor_trq, restated; UNVERIFIED-DEP] */
static double clampd(double x, double lo, double hi){ return x < lo ? lo : ( x > hi ? hi : x ); }
static double motor_inertia(const rkfdModel *m, int i)
{
  return m->mtype[i] == RKFD_MOTOR_DC ? m->mot_inertia[i]*m->mot_gear[i]*m->mot_gear[i] : 0.0;
}
static double motor_input_trq(const rkfdModel *m, int i, double in)
{
  switch( m->mtype[i] ){
  case RKFD_MOTOR_DC:  return m->mot_admit[i]*m->mot_gear[i]*m->mot_k[i]*clampd( in, m->mot_vmin[i], m->mot_vmax[i] );
  case RKFD_MOTOR_TRQ: return clampd( in, m->mot_vmin[i], m->mot_vmax[i] );
  default: return 0.0;
  }
}
static double motor_registance(const rkfdModel *m, int i, double qd)
{
  double gk = m->mot_gear[i]*m->mot_k[i];
  return m->mtype[i] == RKFD_MOTOR_DC ? m->mot_admit[i]*gk*gk*qd : 0.0;
}
static double sgn(double x){ return x > 0 ? 1.0 : ( x < 0 ? -1.0 : 0.0 ); }

/* rkFDKineticFrictionWeight (reference src/rkfd_util.c:193-196) */
static double kf_weight(double w, double fs){ return 1.0 - exp( -1.0*w*fs ); }

/* rkFDJointFriction / rkFDJointFrictionRevolDC (reference src/rkfd_util.c:318-387) */
static void joint_friction(rkfdOracle *o, int doUpRef)
{
  const rkfdModel *m = o->m;
  int i;
  for( i=0; i<o->nl; i++ ){
    Link *l = &o->lk[i];
    int dof = rkfd_joint_dof( m->jtype[i] );
    if( dof != 1 ) continue; /* rkFDJointFrictionAll on float/fixed joints: their kinetic friction is zero */
    if( m->mtype[i] != RKFD_MOTOR_DC ) continue; /* 1-DoF non-DC joints: friction is left untouched (0) */
    {
      double tf, fmax;
      tf = motor_inertia( m, i ) * ( -l->qd / m->dt );
      tf -= motor_input_trq( m, i, o->motor_in[i] );
      tf += motor_registance( m, i, l->qd );
      tf += o->piv_prev[i];
      if( o->piv_type[i] == RKFD_SF )
        fmax = m->sfric[i];
      else /* rkJointGetKFriction: rest torque -k q - c qd - coulomb sgn(qd) */
        fmax = -m->stiff[i]*l->q - m->visc[i]*l->qd - m->coulomb[i]*sgn( l->qd );
      fmax = fabs( fmax );
      if( fabs( tf ) > fmax ){
        tf = tf > 0 ? fmax : -fmax;
        if( doUpRef ) o->piv_type[i] = RKFD_KF;
      } else {
        if( doUpRef ) o->piv_type[i] = RKFD_SF;
      }
      l->tf = tf;
    }
  }
}

/* ------------------------------------------------------------------------ */
/* 6-D transforms between a link and its parent, classical accelerations:
 *   X a  = ( Ra'( a_lin + a_ang x pa ), Ra' a_ang )
 *   X' f = ( Ra f_lin, Ra f_ang + pa x ( Ra f_lin ) )                          */
static void xform_acc(const Link *l, const double *ap, double *r)
{
  double t[3], s[3];
  v3_cross( ap+3, l->pa, t ); v3_add( ap, t, s );
  m3_tmulv( l->Ra, s, r );
  m3_tmulv( l->Ra, ap+3, r+3 );
}
static void xform_force_T(const Link *l, const double *f, double *r)
{
  double fl[3], fa[3], t[3];
  m3_mulv( l->Ra, f, fl ); m3_mulv( l->Ra, f+3, fa );
  v3_cross( l->pa, fl, t );
  v3_copy( fl, r ); v3_add( fa, t, r+3 );
}
static void m6_mulv(const double *M, const double *v, double *r)
{
  int i, j; double t[6];
  for( i=0; i<6; i++ ){ t[i] = 0; for( j=0; j<6; j++ ) t[i] += M[6*i+j]*v[j]; }
  memcpy( r, t, sizeof(t) );
}
/* Y += X' A X */
static void congruence_add(const Link *l, const double *A, double *Y)
{
  double X[36], T[36];
  int i, j, k;
  memset( X, 0, sizeof(X) );
  /* X = [ Ra'  -Ra'[pa]x ; 0  Ra' ] */
  {
    const double *R = l->Ra, *p = l->pa;
    double px[9] = { 0,-p[2],p[1], p[2],0,-p[0], -p[1],p[0],0 }, Rt[9], B[9];
    for( i=0; i<3; i++ ) for( j=0; j<3; j++ ) Rt[3*i+j] = R[3*j+i];
    m3_mul( Rt, px, B );
    for( i=0; i<3; i++ ) for( j=0; j<3; j++ ){
      X[6*i+j] = Rt[3*i+j]; X[6*i+3+j] = -B[3*i+j]; X[6*(3+i)+3+j] = Rt[3*i+j];
    }
  }
  for( i=0; i<6; i++ ) for( j=0; j<6; j++ ){
    double s = 0; for( k=0; k<6; k++ ) s += A[6*i+k]*X[6*k+j];
    T[6*i+j] = s;
  }
  for( i=0; i<6; i++ ) for( j=0; j<6; j++ ){
    double s = 0; for( k=0; k<6; k++ ) s += X[6*k+i]*T[6*k+j];
    Y[6*i+j] += s;
  }
}
/* Cholesky A = L L' (6x6), and solve */
static void chol6(const double *A, double *L)
{
  int i, j, k;
  memset( L, 0, sizeof(double)*36 );
  for( j=0; j<6; j++ ){
    double s = A[6*j+j];
    for( k=0; k<j; k++ ) s -= L[6*j+k]*L[6*j+k];
    L[6*j+j] = sqrt( s );
    for( i=j+1; i<6; i++ ){
      s = A[6*i+j];
      for( k=0; k<j; k++ ) s -= L[6*i+k]*L[6*j+k];
      L[6*i+j] = s / L[6*j+j];
    }
  }
}
static void chol6_solve(const double *L, const double *b, double *x)
{
  int i, k; double y[6];
  for( i=0; i<6; i++ ){ double s = b[i]; for( k=0; k<i; k++ ) s -= L[6*i+k]*y[k]; y[i] = s / L[6*i+i]; }
  for( i=5; i>=0; i-- ){ double s = y[i]; for( k=i+1; k<6; k++ ) s -= L[6*k+i]*x[k]; x[i] = s / L[6*i+i]; }
}

/* velocity-dependent bias + gravity, joint torques: the per-link inputs of the ABA */
static void aba_prepare(rkfdOracle *o)
{
  const rkfdModel *m = o->m;
  int i;
  for( i=0; i<o->nl; i++ ){
    Link *l = &o->lk[i];
    const double *c = &m->com[3*i];
    double ms = m->mass[i], *b = &o->beta0[6*i];
    double wc[3], wwc[3], Iw[3], wIw[3], g[3] = { 0, 0, -RKFD_G }, fg[3], ng[3];
    const double *w = l->v+3;
    /* beta = ( m w x ( w x c ), w x ( Io w ) ) - gravity wrench */
    v3_cross( w, c, wc ); v3_cross( w, wc, wwc );
    Iw[0] = l->M6[6*3+3]*w[0] + l->M6[6*3+4]*w[1] + l->M6[6*3+5]*w[2];
    Iw[1] = l->M6[6*4+3]*w[0] + l->M6[6*4+4]*w[1] + l->M6[6*4+5]*w[2];
    Iw[2] = l->M6[6*5+3]*w[0] + l->M6[6*5+4]*w[1] + l->M6[6*5+5]*w[2];
    v3_cross( w, Iw, wIw );
    m3_tmulv( l->R, g, fg ); v3_mul( fg, ms, fg );
    v3_cross( c, fg, ng );
    b[0] = ms*wwc[0] - fg[0]; b[1] = ms*wwc[1] - fg[1]; b[2] = ms*wwc[2] - fg[2];
    b[3] = wIw[0] - ng[0];    b[4] = wIw[1] - ng[1];    b[5] = wIw[2] - ng[2];
    if( rkfd_joint_dof( m->jtype[i] ) == 1 ){
      l->jm  = motor_inertia( m, i );
      l->tau = motor_input_trq( m, i, o->motor_in[i] ) - motor_registance( m, i, l->qd ) + l->tf;
    }
  }
}

/* bias recursion for one link given beta0, ext and csum: pA, u, contribution to the parent */
static void aba_bias_link(rkfdOracle *o, int i, double *newcontrib)
{
  const rkfdModel *m = o->m;
  Link *l = &o->lk[i];
  double *pA = &o->pA[6*i], *u = &o->u[6*i], pa[6], t[6];
  int k;
  for( k=0; k<6; k++ ) pA[k] = o->beta0[6*i+k] - o->ext[6*i+k] + o->csum[6*i+k];
  switch( ejt( o, i ) ){
  case RKFD_JOINT_REVOL: case RKFD_JOINT_PRISM: {
    int ax = m->jtype[i] == RKFD_JOINT_REVOL ? 5 : 2;
    double ud;
    u[0] = l->tau - pA[ax];
    ud = u[0] / l->D;
    /* pa = pA + Ia gam + U u / D, Ia = IA - U U'/D */
    m6_mulv( l->IA, l->gam, t );
    {
      double ug = 0; for( k=0; k<6; k++ ) ug += l->U[k]*l->gam[k];
      for( k=0; k<6; k++ ) pa[k] = pA[k] + t[k] - l->U[k]*( ug/l->D ) + l->U[k]*ud;
    }
  } break;
  case RKFD_JOINT_FLOAT:
    /* a free joint transmits only its own generalized force (zero) */
    for( k=0; k<6; k++ ){ u[k] = 0; pa[k] = 0; }
    break;
  case RKFD_JOINT_SPHER: {
    /* u = - S' pA (no joint torque), pa = pA + Ia gam + U D^-1 u with Ia = IA - U D^-1 U' */
    double ug[3], w3[3], z3[3];
    int a_, b_;
    for( a_=0; a_<3; a_++ ){
      u[a_] = 0; ug[a_] = 0;
      for( k=0; k<6; k++ ){ u[a_] -= l->S3[3*k+a_]*pA[k]; ug[a_] += l->U3[3*k+a_]*l->gam[k]; }
    }
    for( a_=0; a_<3; a_++ ){
      w3[a_] = 0; z3[a_] = 0;
      for( b_=0; b_<3; b_++ ){ w3[a_] += l->Di3[3*a_+b_]*u[b_]; z3[a_] += l->Di3[3*a_+b_]*ug[b_]; }
    }
    m6_mulv( l->IA, l->gam, t );
    for( k=0; k<6; k++ ){
      pa[k] = pA[k] + t[k];
      for( a_=0; a_<3; a_++ ) pa[k] += l->U3[3*k+a_]*( w3[a_] - z3[a_] );
    }
  } break;
  default:
    m6_mulv( l->IA, l->gam, t );
    for( k=0; k<6; k++ ) pa[k] = pA[k] + t[k];
  }
  xform_force_T( l, pa, newcontrib );
}

/* rkChainUpdateABI [RoKi rk_abi, restated]: sweep 2 (articulated inertia + bias, leaf to
 * root) and sweep 3 (accelerations, root to leaf).  Call sites: reference
 * src/rkfd_sim.c:509-521, src/rkfd_util.c:156 */
static void aba_backward_full(rkfdOracle *o)
{
  const rkfdModel *m = o->m;
  int i, k, j;
  for( i=0; i<o->nl; i++ ){
    memcpy( o->lk[i].IA, o->lk[i].M6, sizeof(double)*36 );
    memset( &o->csum[6*i], 0, sizeof(double)*6 );
  }
  for( i=o->nl-1; i>=0; i-- ){
    Link *l = &o->lk[i];
    double Ia[36], nc[6];
    int par = m->parent[i];
    switch( ejt( o, i ) ){
    case RKFD_JOINT_REVOL: case RKFD_JOINT_PRISM: {
      int ax = m->jtype[i] == RKFD_JOINT_REVOL ? 5 : 2;
      for( k=0; k<6; k++ ) l->U[k] = l->IA[6*k+ax];
      l->D = l->IA[6*ax+ax] + l->jm;
      for( k=0; k<6; k++ ) for( j=0; j<6; j++ ) Ia[6*k+j] = l->IA[6*k+j] - l->U[k]*l->U[j]/l->D;
    } break;
    case RKFD_JOINT_FLOAT:
      chol6( l->IA, l->L6 );
      memset( Ia, 0, sizeof(Ia) );
      break;
    case RKFD_JOINT_SPHER: {
      double D3[9], det;
      int a_, b_;
      for( k=0; k<6; k++ ) for( a_=0; a_<3; a_++ ){
        l->U3[3*k+a_] = 0;
        for( j=0; j<6; j++ ) l->U3[3*k+a_] += l->IA[6*k+j]*l->S3[3*j+a_];
      }
      for( a_=0; a_<3; a_++ ) for( b_=0; b_<3; b_++ ){
        D3[3*a_+b_] = 0;
        for( k=0; k<6; k++ ) D3[3*a_+b_] += l->S3[3*k+a_]*l->U3[3*k+b_];
      }
      det = D3[0]*( D3[4]*D3[8]-D3[5]*D3[7] ) - D3[1]*( D3[3]*D3[8]-D3[5]*D3[6] ) + D3[2]*( D3[3]*D3[7]-D3[4]*D3[6] );
      l->Di3[0] = ( D3[4]*D3[8]-D3[5]*D3[7] )/det; l->Di3[1] = ( D3[2]*D3[7]-D3[1]*D3[8] )/det; l->Di3[2] = ( D3[1]*D3[5]-D3[2]*D3[4] )/det;
      l->Di3[3] = ( D3[5]*D3[6]-D3[3]*D3[8] )/det; l->Di3[4] = ( D3[0]*D3[8]-D3[2]*D3[6] )/det; l->Di3[5] = ( D3[2]*D3[3]-D3[0]*D3[5] )/det;
      l->Di3[6] = ( D3[3]*D3[7]-D3[4]*D3[6] )/det; l->Di3[7] = ( D3[1]*D3[6]-D3[0]*D3[7] )/det; l->Di3[8] = ( D3[0]*D3[4]-D3[1]*D3[3] )/det;
      for( k=0; k<6; k++ ) for( j=0; j<6; j++ ){
        double s_ = 0;
        for( a_=0; a_<3; a_++ ) for( b_=0; b_<3; b_++ ) s_ += l->U3[3*k+a_]*l->Di3[3*a_+b_]*l->U3[3*j+b_];
        Ia[6*k+j] = l->IA[6*k+j] - s_;
      }
    } break;
    default:
      memcpy( Ia, l->IA, sizeof(Ia) );
    }
    aba_bias_link( o, i, nc );
    memcpy( &o->contrib[6*i], nc, sizeof(nc) );
    if( par >= 0 ){
      if( ejt( o, i ) != RKFD_JOINT_FLOAT ) congruence_add( l, Ia, o->lk[par].IA );
      for( k=0; k<6; k++ ) o->csum[6*par+k] += nc[k];
    }
  }
}

/* re-propagate the bias from link i to its root after ext[i] changed
 * (the bias-only part of rkChainUpdateCachedABI / ...CachedABIPair) */
static void aba_bias_path(rkfdOracle *o, int i)
{
  const rkfdModel *m = o->m;
  double nc[6]; int k;
  while( i >= 0 ){
    int par = m->parent[i];
    aba_bias_link( o, i, nc );
    if( par >= 0 )
      for( k=0; k<6; k++ ) o->csum[6*par+k] += nc[k] - o->contrib[6*i+k];
    memcpy( &o->contrib[6*i], nc, sizeof(nc) );
    i = par;
  }
}

static void aba_forward(rkfdOracle *o, double *acc)
{
  const rkfdModel *m = o->m;
  static const double zero6[6] = { 0,0,0,0,0,0 };
  int i, k;
  for( i=0; i<o->nl; i++ ){
    Link *l = &o->lk[i];
    int par = m->parent[i];
    double y[6];
    xform_acc( l, par < 0 ? zero6 : o->lk[par].a, y );
    for( k=0; k<6; k++ ) y[k] += l->gam[k];
    switch( ejt( o, i ) ){
    case RKFD_JOINT_REVOL: case RKFD_JOINT_PRISM: {
      int ax = m->jtype[i] == RKFD_JOINT_REVOL ? 5 : 2;
      double uy = 0, qdd;
      for( k=0; k<6; k++ ) uy += l->U[k]*y[k];
      qdd = ( o->u[6*i] - uy ) / l->D;
      memcpy( l->a, y, sizeof(y) );
      l->a[ax] += qdd;
      acc[m->dofoff[i]] = qdd;
    } break;
    case RKFD_JOINT_FLOAT: {
      double rhs[6], d[6];
      for( k=0; k<6; k++ ) rhs[k] = -o->pA[6*i+k];
      chol6_solve( l->L6, rhs, l->a );
      for( k=0; k<6; k++ ) d[k] = l->a[k] - y[k];
      m3_mulv( l->Rj, d, acc + m->dofoff[i] );
      m3_mulv( l->Rj, d+3, acc + m->dofoff[i] + 3 );
    } break;
    case RKFD_JOINT_SPHER: {
      double r3[3], qdd[3];
      int a_, b_;
      for( a_=0; a_<3; a_++ ){
        r3[a_] = o->u[6*i+a_];
        for( k=0; k<6; k++ ) r3[a_] -= l->U3[3*k+a_]*y[k];
      }
      for( a_=0; a_<3; a_++ ){ qdd[a_] = 0; for( b_=0; b_<3; b_++ ) qdd[a_] += l->Di3[3*a_+b_]*r3[b_]; }
      memcpy( l->a, y, sizeof(y) );
      for( k=0; k<6; k++ ) for( a_=0; a_<3; a_++ ) l->a[k] += l->S3[3*k+a_]*qdd[a_];
      for( a_=0; a_<3; a_++ ) acc[m->dofoff[i]+a_] = qdd[a_];
    } break;
    default:
      memcpy( l->a, y, sizeof(y) );
      if( m->jtype[i] == RKFD_JOINT_BRFLOAT ) for( k=0; k<6; k++ ) acc[m->dofoff[i]+k] = 0;      /* (unbroken: rigidly attached) */
    }
  }
}

/* the break test of the unbroken breakable float joints, at a committing evaluation, after the accelerations are final: the
 * wrench the joint transmits to its link - articulated inertia x acceleration + articulated bias, which for a rigidly attached
 * subtree is the momentum balance of everything that hangs on the joint, contact and penalty wrenches included - in the link
 * frame about the link origin; beyond either threshold the joint is broken from the next evaluation on.
 * [UNVERIFIED-DEP: where RoKi evaluates it (rkJointUpdateWrench inside rkChainUpdateABIWrench, called by the reference only when
 * doUpRef, src/rkfd_sim.c:511-512), the norms and the strict comparison are this restatement's reading] */
static void break_test(rkfdOracle *o)
{
  const rkfdModel *m = o->m;
  int i;
  for( i=0; i<o->nl; i++ ){
    double w[6];
    if( m->jtype[i] != RKFD_JOINT_BRFLOAT || o->broken[i] ) continue;
    m6_mulv( o->lk[i].IA, o->lk[i].a, w );
    { int k; for( k=0; k<6; k++ ) w[k] += o->pA[6*i+k]; }
    if( v3_norm( w ) > m->brk_f[i] || v3_norm( w+3 ) > m->brk_t[i] ) o->broken[i] = 1;
  }
}

/* rkChainSaveABIAccBias / rkChainRestoreABIAccBiasPair [RoKi]: call sites reference
 * src/rkfd_util.c:157,173-181 */
static void aba_save_bias(rkfdOracle *o)
{
  size_t n6 = sizeof(double)*6*o->nl;
  int k;
  /* the wrenches applied so far stay part of the saved bias; the wrench lists are then
   * emptied (rkFDChainExtWrenchDestroy, reference src/rkfd_util.c:158) */
  for( k=0; k<6*o->nl; k++ ){ o->beta0[k] -= o->ext[k]; o->ext[k] = 0; }
  memcpy( o->s_beta0, o->beta0, n6 ); memcpy( o->s_pA, o->pA, n6 ); memcpy( o->s_u, o->u, n6 );
  memcpy( o->s_contrib, o->contrib, n6 ); memcpy( o->s_csum, o->csum, n6 );
}
static void aba_restore_bias(rkfdOracle *o)
{
  size_t n6 = sizeof(double)*6*o->nl;
  memcpy( o->pA, o->s_pA, n6 ); memcpy( o->u, o->s_u, n6 );
  memcpy( o->contrib, o->s_contrib, n6 ); memcpy( o->csum, o->s_csum, n6 );
  memset( o->ext, 0, n6 );
}

/* ------------------------------------------------------------------------ */
/* point kinematics on a link, world frame */
/* rkFDLinkPointWldVel (reference src/rkfd_util.c:14-24) */
static void link_point_vel(const Link *l, const double *x, double *v)
{
  double vw[3], ww[3], r[3], t[3];
  m3_mulv( l->R, l->v, vw ); m3_mulv( l->R, l->v+3, ww );
  v3_sub( x, l->p, r ); v3_cross( ww, r, t ); v3_add( vw, t, v );
}
/* rkFDLinkPointWldAcc (reference src/rkfd_util.c:92-101) with RoKi's rkLinkPointAcc:
 * a_lin + alpha x p + w x ( w x p ) in the link frame */
static void link_point_acc(const Link *l, const double *x, double *a)
{
  double r[3], vp[3], t[3], s[3], al[3];
  v3_sub( x, l->p, r ); m3_tmulv( l->R, r, vp );
  v3_cross( l->a+3, vp, t );
  v3_cross( l->v+3, vp, s ); v3_cross( l->v+3, s, s );
  al[0] = l->a[0]+t[0]+s[0]; al[1] = l->a[1]+t[1]+s[1]; al[2] = l->a[2]+t[2]+s[2];
  m3_mulv( l->R, al, a );
}
/* the two shapes (collision cells) of candidate j: owner of the vertex, the other one */
static void cand_shapes(const rkfdOracle *o, int j, int *own, int *oth)
{
  const rkfdModel *m = o->m;
  int pr = m->cand_pair[j], sd = m->cand_side[j];
  *own = m->pair_shape[2*pr+sd]; *oth = m->pair_shape[2*pr+1-sd];
}
/* rkFDLinkAddSlideVel (reference src/rkfd_util.c:26-40): a cell in slide mode adds the velocity of a surface
 * running around its slide axis, tangential to the contact normal n, of magnitude slide_vel */
static void slide_dir(const rkfdOracle *o, int shape, const double *p, const double *n, double *sv)
{
  const rkfdModel *m = o->m;
  const Link *l = &o->lk[m->shape_link[shape]];
  double tmpv[3], ax[3];
  v3_sub( p, l->p, tmpv );
  m3_mulv( l->R, &m->shape_slide_axis[3*shape], ax );
  v3_cross( ax, tmpv, sv );
  v3_cat( sv, -v3_dot( sv, n ), n );
}
static void add_slide_vel(const rkfdOracle *o, int shape, const double *p, const double *n, double *v)
{
  double sv[3], nr;
  slide_dir( o, shape, p, n, sv );
  nr = v3_norm( sv );
  if( is_tiny( nr ) ) return;
  v3_cat( v, o->m->shape_slide_vel[shape]/nr, sv );
}
/* rkFDChainPointRelativeVel / ...Acc (reference src/rkfd_util.c:42-60,103-118): owner minus other;
 * every cell is RK_CD_CELL_MOVE (reference src/rkfd_sim.c:198) */
static void rel_vel(const rkfdOracle *o, int j, double *v)
{
  double a[3], b[3];
  int own, oth;
  link_point_vel( &o->lk[o->clinkA[j]], &o->cx[3*j], a );
  link_point_vel( &o->lk[o->clinkB[j]], &o->cx[3*j], b );
  cand_shapes( o, j, &own, &oth );
  if( o->m->shape_slide_mode[own] ) add_slide_vel( o, own, &o->cx[3*j], &o->cnorm[3*j], a );
  if( o->m->shape_slide_mode[oth] ) add_slide_vel( o, oth, &o->cx[3*j], &o->cnorm[3*j], b );
  v3_sub( a, b, v );
}
/* rkFDUpdateRefSlide (reference src/rkfd_util.c:218-237): when a vertex stays in static friction on a cell in
 * slide mode, its anchor _ref is carried along by dt * slide_vel.  cell[0], cell[1] are the pair's shapes in
 * registration order; the anchor lives in the frame named by the reference's index expression. */
static void update_ref_slide(rkfdOracle *o, int j)
{
  const rkfdModel *m = o->m;
  int pr = m->cand_pair[j], sd = m->cand_side[j], i;
  for( i=0; i<2; i++ ){
    int sh = m->pair_shape[2*pr+i], isown = ( i == sd ), tgt;
    double sv[3], nr, t[3];
    if( !m->shape_slide_mode[sh] ) continue;
    slide_dir( o, sh, &o->cx[3*j], &o->cnorm[3*j], sv );
    nr = v3_norm( sv );
    if( is_tiny( nr ) ) continue;
    v3_mul( sv, ( isown ? -1.0 : 1.0 )*m->dt*m->shape_slide_vel[sh]/nr, sv );
    tgt = m->pair_shape[2*pr+( isown ? 1 : 0 )];
    m3_tmulv( o->lk[m->shape_link[tgt]].R, sv, t );
    v3_add( &o->cv_ref[3*j], t, &o->cv_ref[3*j] );
  }
}
static void rel_acc(const rkfdOracle *o, int j, double *r)
{
  double a[3], b[3];
  link_point_acc( &o->lk[o->clinkA[j]], &o->cx[3*j], a );
  link_point_acc( &o->lk[o->clinkB[j]], &o->cx[3*j], b );
  v3_sub( a, b, r );
}

/* add a world force fw acting at world point x on link i to the link's wrench buffer
 * (link frame, about the link origin); sign = +1 / -1 */
static void ext_add(rkfdOracle *o, int i, const double *x, const double *fw, double sign)
{
  const Link *l = &o->lk[i];
  double r[3], pos[3], f[3], n[3]; int k;
  v3_sub( x, l->p, r ); m3_tmulv( l->R, r, pos );
  m3_tmulv( l->R, fw, f ); v3_mul( f, sign, f );
  v3_cross( pos, f, n );
  for( k=0; k<3; k++ ){ o->ext[6*i+k] += f[k]; o->ext[6*i+3+k] += n[k]; }
}
/* rkFDContactForcePushWrench (reference src/rkfd_util.c:268-282) */
static void push_wrench(rkfdOracle *o, int j)
{
  ext_add( o, o->clinkA[j], &o->cx[3*j], &o->cv_f[3*j],  1.0 );
  ext_add( o, o->clinkB[j], &o->cx[3*j], &o->cv_f[3*j], -1.0 );
}

/* ------------------------------------------------------------------------ */
/* rkCDColChkVert [RoKi rk_cd, restated for convex shapes; UNVERIFIED-DEP] followed by
 * rkFDCDUpdate (reference src/rkfd_cd.c:33-49): a candidate vertex of one shape is in contact
 * when it lies inside the other shape; the contact normal is the outward normal of the face
 * of least penetration; the stick anchor _ref (other link's frame) is set at first contact. */
static void collision(rkfdOracle *o)
{
  const rkfdModel *m = o->m;
  int j, f;
  o->nel = o->nrg = 0;
  for( j=0; j<o->ncand; j++ ){
    int pr = m->cand_pair[j], sd = m->cand_side[j];
    int shA = m->pair_shape[2*pr+sd], shB = m->pair_shape[2*pr+1-sd];
    int la = m->shape_link[shA], lb = m->shape_link[shB];
    const Link *A = &o->lk[la], *B = &o->lk[lb];
    double x[3], y[3], r[3], smax = -HUGE_VAL; int fbest = -1;
    m3_mulv( A->R, &m->verts[3*m->cand_vert[j]], x ); v3_add( x, A->p, x );
    v3_sub( x, B->p, r ); m3_tmulv( B->R, r, y );
    for( f=m->shape_foff[shB]; f<m->shape_foff[shB+1]; f++ ){
      double s = v3_dot( &m->planes[4*f], y ) - m->planes[4*f+3];
      if( s > smax ){ smax = s; fbest = f; }
    }
    o->clinkA[j] = la; o->clinkB[j] = lb; o->cci[j] = m->pair_ci[pr];
    v3_copy( x, &o->cx[3*j] );
    v3_zero( &o->cv_f[3*j] );
    if( fbest < 0 || !( smax < TOL ) ){
      o->cv_active[j] = 0;
      continue;
    }
    {
      const double *pl = &m->planes[4*fbest];
      double pro[3];
      v3_copy( y, pro ); v3_cat( pro, -smax, pl );
      v3_copy( pro, &o->cpro[3*j] );
      m3_mulv( B->R, pl, &o->cnorm[3*j] );
      if( !o->cv_active[j] ){
        o->cv_active[j] = 1;
        o->cv_type[j] = RKFD_SF;
        v3_copy( pro, &o->cv_ref[3*j] );
      }
      m3_mulv( B->R, &o->cv_ref[3*j], &o->crefw[3*j] ); v3_add( &o->crefw[3*j], B->p, &o->crefw[3*j] );
      v3_copy( &o->cnorm[3*j], &o->caxis[9*j] );
      ortho_space( &o->cnorm[3*j], &o->caxis[9*j+3], &o->caxis[9*j+6] );
    }
    if( m->ci_type[o->cci[j]] == RKFD_CONTACT_ELASTIC ) o->el[o->nel++] = j;
    else if( m->ci_type[o->cci[j]] == RKFD_CONTACT_RIGID ) o->rg[o->nrg++] = j;
  }
}

/* rkFDContactForceModifyFriction (reference src/rkfd_util.c:239-266) */
static void modify_friction(rkfdOracle *o, int j, const double *vr, int doUpRef)
{
  const rkfdModel *m = o->m;
  double *f = &o->cv_f[3*j], *ax = &o->caxis[9*j], v[3];
  double fn, fs, vs, mu;
  int ci = o->cci[j];
  fn = v3_dot( f, ax );
  fs = sqrt( v3_dot( f, ax+3 )*v3_dot( f, ax+3 ) + v3_dot( f, ax+6 )*v3_dot( f, ax+6 ) );
  mu = o->cv_type[j] == RKFD_SF ? m->ci_sf[ci] : m->ci_kf[ci];
  if( !is_tiny( fs ) && fs > mu*fn ){
    v3_copy( vr, v );
    v3_cat( v, -v3_dot( v, ax ), ax );
    vs = v3_norm( v );
    v3_mul( ax, fn, f );
    if( !is_tiny( vs ) ){
      v3_mul( v, 1.0/vs, v );
      v3_cat( f, -kf_weight( m->friction_weight, vs )*m->ci_kf[ci]*fn, v );
    }
    if( doUpRef ){
      o->cv_type[j] = RKFD_KF;
      v3_copy( &o->cpro[3*j], &o->cv_ref[3*j] );
    }
  } else {
    if( doUpRef ){ o->cv_type[j] = RKFD_SF; update_ref_slide( o, j ); }
  }
}

/* rkFDSolverPenalty (reference src/rkfd_penalty.c:11-31) */
static void penalty(rkfdOracle *o, int doUpRef)
{
  const rkfdModel *m = o->m;
  int e;
  for( e=0; e<o->nel; e++ ){
    int j = o->el[e], ci = o->cci[j];
    double d[3], vr[3], *f = &o->cv_f[3*j];
    v3_sub( &o->cx[3*j], &o->crefw[3*j], d );
    rel_vel( o, j, vr );
    v3_mul( d, -m->ci_e[ci], f );
    v3_cat( f, -1.0*( m->ci_v[ci] + m->ci_e[ci]*m->dt ), vr );
    if( v3_dot( f, &o->caxis[9*j] ) < 0.0 ){ v3_zero( f ); continue; }
    modify_friction( o, j, vr, doUpRef );
    push_wrench( o, j );
  }
}

/* ------------------------------------------------------------------------ */
/* MLCP plugin, rigid branch: _rkFDSolverConstraint (reference src/rkfd_mlcp.c:287-297) */
/* _rkFDSolverRelationAccForce (reference src/rkfd_mlcp.c:124-142 = src/rkfd_vert.c:153-189): the
 * free relative accelerations b and, column by column, the response matrix A of the rigid contact
 * vertices to unit forces along their axes.  Shared by the MLCP and the Vert plugin. */
static void contact_system(rkfdOracle *o)
{
  const rkfdModel *m = o->m;
  int nc = o->nrg, n3 = 3*nc, c, i, r;
  double *A = o->ma, *b = o->mb, *t = o->mt;

  /* rkFDUpdateAccBias (reference src/rkfd_util.c:149-161): full ABA, save bias, drop wrenches */
  aba_backward_full( o );
  aba_forward( o, o->acc );
  aba_save_bias( o );
  /* _rkFDSolverBiasAcc (reference src/rkfd_mlcp.c:58-74) */
  for( c=0; c<nc; c++ ){
    double av[3];
    rel_acc( o, o->rg[c], av );
    for( i=0; i<3; i++ ) b[3*c+i] = v3_dot( &o->caxis[9*o->rg[c]+3*i], av );
  }
  /* probe every contact axis with a unit wrench (reference src/rkfd_mlcp.c:104-122,135-141) */
  for( c=0; c<nc; c++ ){
    int j = o->rg[c];
    for( i=0; i<3; i++ ){
      const double *axis = &o->caxis[9*j+3*i];
      ext_add( o, o->clinkA[j], &o->cx[3*j], axis,  1.0 );
      ext_add( o, o->clinkB[j], &o->cx[3*j], axis, -1.0 );
      aba_bias_path( o, o->clinkA[j] );
      aba_bias_path( o, o->clinkB[j] );
      aba_forward( o, o->acc );
      /* _rkFDSolverRelativeAcc (reference src/rkfd_mlcp.c:76-102) */
      for( r=0; r<nc; r++ ){
        int jr = o->rg[r], k;
        int chA = m->chain[o->clinkA[jr]], chB = m->chain[o->clinkB[jr]];
        int pA = m->chain[o->clinkA[j]], pB = m->chain[o->clinkB[j]];
        if( chA != pA && chA != pB && chB != pA && chB != pB ){
          for( k=0; k<3; k++ ) t[3*r+k] = 0;
        } else {
          double av[3];
          rel_acc( o, jr, av );
          for( k=0; k<3; k++ ) t[3*r+k] = v3_dot( &o->caxis[9*jr+3*k], av ) - b[3*r+k];
        }
      }
      for( r=0; r<n3; r++ ) A[n3*r+3*c+i] = t[r];
      aba_restore_bias( o );
    }
  }
}

static int mlcp(rkfdOracle *o, int doUpRef)
{
  const rkfdModel *m = o->m;
  int nc = o->nrg, n3 = 3*nc, c, i, r, cnt;
  double *A = o->ma, *b = o->mb, *f = o->mf;
  double dt = m->dt;
  (void)doUpRef;

  o->last_nc = nc;
  contact_system( o );
  /* _rkFDSolverBiasVel (reference src/rkfd_mlcp.c:146-162) */
  for( r=0; r<n3; r++ ) b[r] *= dt;
  for( c=0; c<nc; c++ ){
    int j = o->rg[c];
    rel_vel( o, j, &o->cvel[3*j] );
    for( i=0; i<3; i++ ) b[3*c+i] += v3_dot( &o->cvel[3*j], &o->caxis[9*j+3*i] );
  }
  /* _rkFDSolverRelaxationCompensation (reference src/rkfd_mlcp.c:164-188) */
  for( c=0; c<nc; c++ ){
    int j = o->rg[c], ci = o->cci[j];
    double d[3], k;
    v3_sub( &o->cx[3*j], &o->crefw[3*j], d );
    for( i=0; i<3; i++ ) A[n3*(3*c+i)+3*c+i] += m->ci_l[ci];
    k = o->cv_type[j] == RKFD_SF ? m->ci_sf[ci] : m->ci_kf[ci];
    b[3*c  ] += m->ci_k[ci]     * v3_dot( d, &o->caxis[9*j] );
    b[3*c+1] += m->ci_k[ci] * k * v3_dot( d, &o->caxis[9*j+3] );
    b[3*c+2] += m->ci_k[ci] * k * v3_dot( d, &o->caxis[9*j+6] );
  }
  /* _rkFDSolverMLCP (reference src/rkfd_mlcp.c:190-249): fixed max_iter sweeps, no warm start */
  for( r=0; r<n3; r++ ) f[r] = 0;
  for( cnt=0; cnt<m->max_iter; cnt++ ){
    for( c=0; c<nc; c++ ){
      int off = 3*c; double dot = 0, ff;
      for( r=0; r<n3; r++ ) dot += A[n3*off+r]*f[r];
      ff = -( b[off] + dot - A[n3*off+off]*f[off] ) / A[n3*off+off];
      f[off] = ff < TOL ? 0.0 : ff;
    }
    for( c=0; c<nc; c++ ){
      int off = 3*c, j = o->rg[c], ci = o->cci[j];
      double ff[2], fnorm, fs;
      for( i=0; i<2; i++ ){
        int ro = off+1+i;
        if( fabs( A[n3*ro+ro] ) < TOL ) ff[i] = 0;
        else {
          double dot = 0;
          for( r=0; r<n3; r++ ) dot += A[n3*ro+r]*f[r];
          ff[i] = -( b[ro] + dot - A[n3*ro+ro]*f[ro] ) / A[n3*ro+ro];
        }
      }
      fnorm = ff[0]*ff[0] + ff[1]*ff[1];
      fs = o->cv_type[j] == RKFD_SF ? m->ci_sf[ci]*f[off] : m->ci_kf[ci]*f[off];
      fs = fs*fs;
      if( fnorm < TOL || fs < TOL ){
        f[off+1] = 0.0; f[off+2] = 0.0;
      } else if( fnorm > fs ){
        fs /= fnorm;
        f[off+1] = ff[0]*fs; f[off+2] = ff[1]*fs;
      } else {
        f[off+1] = ff[0]; f[off+2] = ff[1];
      }
    }
  }
  for( r=0; r<n3; r++ ) f[r] /= dt;
  /* _rkFDSolverSetForce (reference src/rkfd_mlcp.c:252-284); quirks Q1 (world components
   * e[0] / e[1],e[2] used as normal / tangential) and Q2 (state updated on every evaluation) */
  for( c=0; c<nc; c++ ){
    int j = o->rg[c], ci = o->cci[j];
    double *fw = &o->cv_f[3*j], fn, fs, mu;
    v3_zero( fw );
    for( i=0; i<3; i++ ) v3_cat( fw, f[3*c+i], &o->caxis[9*j+3*i] );
    push_wrench( o, j );
    fn = fw[0];
    fs = sqrt( fw[1]*fw[1] + fw[2]*fw[2] );
    mu = o->cv_type[j] == RKFD_SF ? m->ci_sf[ci] : m->ci_kf[ci];
    if( fs > mu*fn - TOL ){
      o->cv_type[j] = RKFD_KF;
      v3_copy( &o->cpro[3*j], &o->cv_ref[3*j] );
    } else {
      o->cv_type[j] = RKFD_SF;
      update_ref_slide( o, j );      /* reference src/rkfd_mlcp.c:279 */
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------ */
/* Vert plugin, rigid branch (reference src/rkfd_vert.c:258-336) with its active-set QP solver
 * (reference src/rkfd_opt_qp.c:43-181).
 *
 * [UNVERIFIED-DEP] zLESolveMP( K, rhs, NULL, NULL, x ) (ZM, un-vendored; reference
 * src/rkfd_opt_qp.c:107) is taken to return the Moore-Penrose solution x = K^+ rhs (minimum-norm
 * least-squares solution with unit weights).  The KKT matrix K is symmetric, so K^+ is formed here
 * from a cyclic-Jacobi eigen-decomposition, eigenvalues below 1e-12 |lambda|_max counting as zero
 * (the rank tolerance is part of the unverifiable dependency).  zQuadraticValue(q,c,x) is taken as
 * x'Qx/2 + c'x, zEqual(a,b,tol) as |a-b| < tol. */
static void sym_pinv_solve(int n, double *K, const double *rhs, double *x)
{
  /* K (n x n, symmetric, destroyed) = V diag(w) V';  x = V diag(1/w) V' rhs over the non-zero w */
  double *V = (double *)malloc( sizeof(double)*n*n ), *y = (double *)malloc( sizeof(double)*n );
  int i, j, k, sweep;
  double wmax = 0;
  for( i=0; i<n; i++ ) for( j=0; j<n; j++ ) V[n*i+j] = i == j ? 1.0 : 0.0;
  for( sweep=0; sweep<60; sweep++ ){
    double off = 0, diag = 0;
    for( i=0; i<n; i++ ){ diag += K[n*i+i]*K[n*i+i]; for( j=i+1; j<n; j++ ) off += K[n*i+j]*K[n*i+j]; }
    if( off <= 1e-30*( diag + off ) || off == 0 ) break;
    for( i=0; i<n-1; i++ )
      for( j=i+1; j<n; j++ ){
        double apq = K[n*i+j], theta, t, c, sn;
        if( apq == 0.0 ) continue;
        theta = ( K[n*j+j] - K[n*i+i] )/( 2.0*apq );
        t = ( theta >= 0 ? 1.0 : -1.0 )/( fabs( theta ) + sqrt( theta*theta + 1.0 ) );
        c = 1.0/sqrt( t*t + 1.0 ); sn = t*c;
        for( k=0; k<n; k++ ){
          double kp = K[n*k+i], kq = K[n*k+j];
          K[n*k+i] = c*kp - sn*kq; K[n*k+j] = sn*kp + c*kq;
        }
        for( k=0; k<n; k++ ){
          double kp = K[n*i+k], kq = K[n*j+k];
          K[n*i+k] = c*kp - sn*kq; K[n*j+k] = sn*kp + c*kq;
        }
        for( k=0; k<n; k++ ){
          double vp = V[n*k+i], vq = V[n*k+j];
          V[n*k+i] = c*vp - sn*vq; V[n*k+j] = sn*vp + c*vq;
        }
      }
  }
  for( i=0; i<n; i++ ) if( fabs( K[n*i+i] ) > wmax ) wmax = fabs( K[n*i+i] );
  for( i=0; i<n; i++ ){
    double s = 0;
    for( k=0; k<n; k++ ) s += V[n*k+i]*rhs[k];
    y[i] = fabs( K[n*i+i] ) > 1e-12*wmax ? s/K[n*i+i] : 0.0;
  }
  for( k=0; k<n; k++ ){
    double s = 0;
    for( i=0; i<n; i++ ) s += V[n*k+i]*y[i];
    x[k] = s;
  }
  free( V ); free( y );
}

/* _rkFDSolverQPASMCond (reference src/rkfd_vert.c:246-250): constraint i touches only the three
 * force components of its contact */
static double qp_cond(const double *nf, int n, int P, const double *ans, int i)
{
  if( P > 0 ){
    int c = i/P;
    return v3_dot( &nf[n*i+3*c], &ans[3*c] );
  } else {
    double s = 0; int j;
    for( j=0; j<n; j++ ) s += nf[n*i+j]*ans[j];
    return s;
  }
}

#define QP_ASM_TOL 1.0e-8
/* rkFDQPSolveASM (reference src/rkfd_opt_qp.c:43-181): min x'Qx/2 + c'x  s.t.  nf x >= d.
 * n unknowns, mc constraints; idx = active-set flags (output as well: used for the stick / slip
 * decision).  Returns the number of KKT solves. */
static int qp_asm_ex(int n, int mc, int P, const double *q, const double *c, const double *nf, const double *d, const double *init, double *ans, int *idx);
static int qp_asm(int n, int mc, int P, const double *q, const double *c, const double *nf, const double *d, double *ans, int *idx)
{
  return qp_asm_ex( n, mc, P, q, c, nf, d, NULL, ans, idx );
}
/* P > 0: the Vert plugin's condition function (a row touches the three unknowns of contact i / P) and start point;
 * P == 0: the solver's defaults for the condition (_rkFDQPSolveASMConditionDefault, :22-25: the whole row) with the
 * caller's start point init (the Volume plugin, reference src/rkfd_volume.c:530-546) */
static int qp_asm_ex(int n, int mc, int P, const double *q, const double *c, const double *nf, const double *d, const double *init, double *ans, int *idx)
{
  int nmax = n + mc, m = 0, nm, i, j, k, iter = 0, nhist = 0, caphist = 16;
  double *qa = (double *)malloc( sizeof(double)*nmax*nmax ), *xy = (double *)malloc( sizeof(double)*nmax );
  double *cb = (double *)malloc( sizeof(double)*nmax ), *dv = (double *)malloc( sizeof(double)*n );
  int *hidx = (int *)malloc( sizeof(int)*mc*caphist );
  double *hmin = (double *)malloc( sizeof(double)*caphist );

  /* _rkFDSolverQPASMInit (reference src/rkfd_vert.c:235-244): unit normal forces */
  for( i=0; i<n; i++ ) ans[i] = 0.0;
  if( init ) for( i=0; i<n; i++ ) ans[i] = init[i];
  else for( i=0; i<n/3; i++ ) ans[3*i] = 1.0;
  /* _rkFDQPSolveASMInitIndex (:29-42) */
  for( i=0; i<mc; i++ ){
    idx[i] = fabs( qp_cond( nf, n, P, ans, i ) - d[i] ) < TOL ? 1 : 0;
    m += idx[i];
  }
  for(;;){
    int step2 = 0;
    double tempd, tempd2, objv;
    nm = n + m;
    for( i=0; i<n; i++ ) for( j=0; j<n; j++ ) qa[nm*i+j] = -q[n*i+j];
    for( k=0, j=n; j<nm; j++ ){
      while( !idx[k] ) k++;
      for( i=0; i<n; i++ ){ qa[nm*i+j] = nf[n*k+i]; qa[nm*j+i] = nf[n*k+i]; }
      k++;
    }
    for( i=n; i<nm; i++ ) for( j=n; j<nm; j++ ) qa[nm*i+j] = 0.0;
    for( i=0; i<n; i++ ) cb[i] = c[i];
    for( k=0, i=n; i<nm; i++ ){ while( !idx[k] ) k++; cb[i] = d[k]; k++; }
    sym_pinv_solve( nm, qa, cb, xy );
    iter++;

    for( i=0; i<n; i++ ) if( !( fabs( xy[i] - ans[i] ) < TOL ) ){ step2 = 1; break; }
    if( getenv( "RKFD_QP_TRACE" ) ){
      unsigned long long mk = 0; for( i=0; i<mc && i<64; i++ ) if( idx[i] ) mk |= 1ull << i;
      printf( "orc it %d mask %016llx m %d moved %d x", iter-1, mk, m, step2 ); for( i=0; i<n; i++ ) printf( " %.6e", xy[i] ); printf( "\n" );
    }
    if( !step2 ){
      int neg = 0, tempi;
      for( i=0; i<n; i++ ) ans[i] = xy[i];
      for( i=0; i<m; i++ ) if( xy[n+i] < 0 ) neg = 1;
      if( !neg ) break;                       /* found the optimal solution */
      tempd = xy[n];
      for( i=1; i<m; i++ ) if( xy[n+i] < tempd ) tempd = xy[n+i];
      tempi = 0;
      for( i=0; i<mc; i++ )
        if( idx[i] ){
          if( fabs( xy[tempi+n] - tempd ) < QP_ASM_TOL ){ idx[i] = 0; m--; }
          tempi++;
        }
      continue;
    }
    /* STEP2: move towards the equality-constrained minimiser as far as feasibility allows */
    for( i=0; i<n; i++ ) dv[i] = xy[i] - ans[i];
    tempd = 1.0;
    for( i=0; i<mc; i++ ){
      tempd2 = 0;
      for( j=0; j<n; j++ ) tempd2 += nf[n*i+j]*dv[j];
      if( idx[i] == 0 && tempd2 < 0 ){
        tempd2 = ( d[i] - qp_cond( nf, n, P, ans, i ) )/tempd2;
        if( tempd2 < tempd ) tempd = tempd2;
      }
    }
    if( getenv( "RKFD_QP_TRACE" ) ) printf( "orc    step t %.12e\n", tempd );
    for( i=0; i<n; i++ ) ans[i] += tempd*dv[i];
    for( i=0; i<mc; i++ )
      if( idx[i] == 0 && fabs( qp_cond( nf, n, P, ans, i ) - d[i] ) < TOL ){ idx[i] = 1; m++; }
    /* check if circulation happens due to degeneracy (:150-161) */
    objv = 0;
    for( i=0; i<n; i++ ){
      double s = 0;
      for( j=0; j<n; j++ ) s += q[n*i+j]*ans[j];
      objv += 0.5*ans[i]*s + c[i]*ans[i];
    }
    {
      int endflag = 0, h;
      for( h=0; h<nhist; h++ ){
        int same = 1;
        for( i=0; i<mc; i++ ) if( idx[i] != hidx[mc*h+i] ){ same = 0; break; }
        if( !same ) continue;
        if( fabs( hmin[h]/objv - 1.0 ) > QP_ASM_TOL ) continue;
        endflag = 1;
        break;
      }
      if( endflag ){ iter = -iter; break; }      /* (negative count: stopped by the circulation check, not at the optimum) */
    }
    if( nhist == caphist ){
      caphist *= 2;
      hidx = (int *)realloc( hidx, sizeof(int)*mc*caphist ); hmin = (double *)realloc( hmin, sizeof(double)*caphist );
    }
    for( i=0; i<mc; i++ ) hidx[mc*nhist+i] = idx[i];
    hmin[nhist++] = objv;
  }
  free( qa ); free( xy ); free( cb ); free( dv ); free( hidx ); free( hmin );
  return iter;
}

static int vert_rigid(rkfdOracle *o, int doUpRef)
{
  const rkfdModel *m = o->m;
  const int nc = o->nrg, n3 = 3*nc, P = m->pyramid > 0 ? m->pyramid : 8, mc = P*nc;
  int c, i, r, k;
  double *A = o->ma, *b = o->mb, *f = o->mf;
  const double dt = m->dt;
  double *q = (double *)malloc( sizeof(double)*n3*n3 ), *cv = (double *)malloc( sizeof(double)*n3 );
  double *nf = (double *)calloc( (size_t)mc*n3, sizeof(double) ), *d = (double *)calloc( mc, sizeof(double) );
  double *cc = (double *)malloc( sizeof(double)*n3 );
  int *idx = (int *)malloc( sizeof(int)*mc );

  o->last_nc = nc;
  /* _rkFDSolverFrictionConstraint (reference src/rkfd_vert.c:73-103); the sine / cosine table of
   * rkFDCrateSinCosTable( table, P, -pi/P ) (src/rkfd_util.c:199-214, src/rkfd_vert.c:367) */
  for( c=0; c<nc; c++ ){
    int j = o->rg[c], ci = o->cci[j];
    const double off = -M_PI/P, dth = 2.0*M_PI/P;
    double fric = o->cv_type[j] == RKFD_KF ? m->ci_kf[ci] : m->ci_sf[ci], th = 0.0;
    fric *= cos( 0.0 + off );
    for( i=0; i<P; i++, th+=dth ){
      nf[n3*( P*c+i )+3*c  ] = fric;
      nf[n3*( P*c+i )+3*c+1] = sin( th + off );
      nf[n3*( P*c+i )+3*c+2] = cos( th + off );
    }
  }
  /* _rkFDSolverRelationAccForce (:153-189) */
  contact_system( o );
  /* _rkFDSolverBiasVel (:193-210) */
  for( r=0; r<n3; r++ ) b[r] *= dt;
  for( c=0; c<nc; c++ ){
    int j = o->rg[c];
    rel_vel( o, j, &o->cvel[3*j] );
    for( i=0; i<3; i++ ) b[3*c+i] += v3_dot( &o->cvel[3*j], &o->caxis[9*j+3*i] );
  }
  /* _rkFDSolverCompensateDepth (:212-233) */
  for( c=0; c<nc; c++ ){
    int j = o->rg[c], ci = o->cci[j];
    double dd[3], fric = o->cv_type[j] == RKFD_KF ? m->ci_kf[ci] : m->ci_sf[ci], kk = m->ci_k[ci];
    v3_sub( &o->cx[3*j], &o->crefw[3*j], dd );
    cc[3*c  ] = b[3*c  ] + kk      * v3_dot( dd, &o->caxis[9*j] );
    cc[3*c+1] = b[3*c+1] + kk*fric * v3_dot( dd, &o->caxis[9*j+3] );
    cc[3*c+2] = b[3*c+2] + kk*fric * v3_dot( dd, &o->caxis[9*j+6] );
  }
  /* _rkFDSolverQP (:258-283): q = A'A + L, c = A'c */
  for( i=0; i<n3; i++ )
    for( k=0; k<n3; k++ ){
      double s = 0;
      for( r=0; r<n3; r++ ) s += A[n3*r+i]*A[n3*r+k];
      q[n3*i+k] = s;
    }
  for( i=0; i<n3; i++ ){
    double s = 0;
    for( r=0; r<n3; r++ ) s += A[n3*r+i]*cc[r];
    cv[i] = s;
  }
  for( c=0; c<nc; c++ ){
    int ci = o->cci[o->rg[c]];
    for( i=0; i<3; i++ ) q[n3*( 3*c+i )+3*c+i] += m->ci_l[ci];
  }
  o->last_qp_iter = qp_asm( n3, mc, P, q, cv, nf, d, f, idx );
  if( o->last_qp_iter < 0 ){ o->last_qp_iter = -o->last_qp_iter; o->qp_cycle_stops++; }
  free( o->qp_q ); free( o->qp_c ); free( o->qp_nf ); free( o->qp_ans ); free( o->qp_idx );
  o->qp_n = n3; o->qp_mc = mc;
  o->qp_q = (double *)malloc( sizeof(double)*n3*n3 ); memcpy( o->qp_q, q, sizeof(double)*n3*n3 );
  o->qp_c = (double *)malloc( sizeof(double)*n3 ); memcpy( o->qp_c, cv, sizeof(double)*n3 );
  o->qp_nf = (double *)malloc( sizeof(double)*mc*n3 ); memcpy( o->qp_nf, nf, sizeof(double)*mc*n3 );
  o->qp_ans = (double *)malloc( sizeof(double)*n3 ); memcpy( o->qp_ans, f, sizeof(double)*n3 );
  o->qp_idx = (int *)malloc( sizeof(int)*mc ); memcpy( o->qp_idx, idx, sizeof(int)*mc );
  for( r=0; r<n3; r++ ) f[r] /= dt;
  /* _rkFDSolverSetForce (:286-323): unlike the MLCP plugin, contact state is committed only when doUpRef */
  for( c=0; c<nc; c++ ){
    int j = o->rg[c];
    double *fw = &o->cv_f[3*j];
    v3_zero( fw );
    for( i=0; i<3; i++ ) v3_cat( fw, f[3*c+i], &o->caxis[9*j+3*i] );
    push_wrench( o, j );
    if( doUpRef ){
      int flag = 0;
      for( i=0; i<P; i++ ) if( idx[P*c+i] ){ flag = 1; break; }
      if( flag ){
        o->cv_type[j] = RKFD_KF;
        v3_copy( &o->cpro[3*j], &o->cv_ref[3*j] );
      } else {
        o->cv_type[j] = RKFD_SF;
        update_ref_slide( o, j );     /* reference src/rkfd_vert.c:318 */
      }
    }
  }
  free( q ); free( cv ); free( nf ); free( d ); free( cc ); free( idx );
  return 0;
}

#include "rkfd_oracle_volume.h"

/* ------------------------------------------------------------------------ */
/* _rkFDUpdate / _rkFDUpdateRef (reference src/rkfd_sim.c:533-549) */
static int evaluate(rkfdOracle *o, const double *dis, const double *vel, double *acc, int doUpRef)
{
  const rkfdModel *m = o->m;
  int i, cached = 0;

  memset( acc, 0, sizeof(double)*o->n );
  o->last_nc = 0;
  connect_state( o, dis, vel );
  /* _rkFDUpdateReset (reference src/rkfd_sim.c:445-453) */
  memset( o->ext, 0, sizeof(double)*6*o->nl );
  for( i=0; i<o->nl; i++ ) o->lk[i].tf = 0; /* friction is re-derived below for DC joints */
  /* _rkFDUpdateCD (reference src/rkfd_sim.c:466-471) */
  collision( o );
  /* rkFDSolverUpdate_<plugin> (reference src/rkfd_mlcp.c:335-343, src/rkfd_vert.c:380-388) */
  joint_friction( o, doUpRef );
  aba_prepare( o );
  if( o->nel != 0 ) penalty( o, doUpRef );
  if( m->solver == RKFD_SOLVER_VOLUME ){
    /* rkFDSolverColChk_Volume / rkFDSolverUpdate_Volume (reference src/rkfd_volume.c:1003-1019): the rigid pairs go by
     * their intersection volumes, not by contact vertices */
    vol_collision( o );
    if( o->nvp != 0 ){ if( volume_rigid( o, doUpRef ) < 0 ) return -1; cached = 1; }
  } else if( o->nrg != 0 ){
    if( m->solver == RKFD_SOLVER_MLCP ){ if( mlcp( o, doUpRef ) < 0 ) return -1; }
    else if( m->solver == RKFD_SOLVER_VERT ){ if( vert_rigid( o, doUpRef ) < 0 ) return -1; }
    else return -1;
    cached = 1;
  }
  /* _rkFDUpdateAcc (reference src/rkfd_sim.c:502-523) */
  if( cached ){
    /* rkChainUpdateCachedABI: saved bias + the wrenches pushed since, then sweep 3 */
    for( i=o->nl-1; i>=0; i-- ){
      int k, any = 0;
      for( k=0; k<6; k++ ) if( o->ext[6*i+k] != 0.0 ) any = 1;
      if( any ) aba_bias_path( o, i );
    }
    aba_forward( o, acc );
  } else {
    aba_backward_full( o );
    aba_forward( o, acc );
  }
  if( doUpRef ) break_test( o );
  return 0;
}

/* rkFDUpdateJointPrevDrivingTrq (reference src/rkfd_util.c:289-311) */
static void update_prev_trq(rkfdOracle *o)
{
  const rkfdModel *m = o->m;
  int i;
  for( i=0; i<o->nl; i++ ){
    if( rkfd_joint_dof( m->jtype[i] ) != 1 ) continue;
    {
      Link *l = &o->lk[i];
      double drv = motor_input_trq( m, i, o->motor_in[i] ) - motor_registance( m, i, l->qd )
                 - motor_inertia( m, i ) * o->acc[m->dofoff[i]];
      o->piv_prev[i] = drv + l->tf;
    }
  }
}

static int eval_ref(rkfdOracle *o)
{
  int r = evaluate( o, o->dis, o->vel, o->acc, 1 );
  update_prev_trq( o );
  return r;
}

int rkfdOracleEval(rkfdOracle *o, int doUpRef)
{
  if( doUpRef ) return eval_ref( o );
  return evaluate( o, o->dis, o->vel, o->acc, 0 );
}

void rkfdOracleUpdateInit(rkfdOracle *o){ eval_ref( o ); }

/* rkFDODECatDefault (reference src/rkfd_sim.c:306-320) with RoKi's rkChainCatJointDisAll:
 * xnew = x (+) k v; float joints compose the rotation R(k w) R(aa)  [UNVERIFIED-DEP] */
static void cat_dis(const rkfdOracle *o, const double *x, double k, const double *v, double *xn)
{
  const rkfdModel *m = o->m;
  int i, j;
  for( i=0; i<o->nl; i++ ){
    int off = m->dofoff[i];
    switch( kjt( o, i ) ){
    case RKFD_JOINT_FLOAT: {
      double aa[3], Rk[9], R0[9], Rn[9];
      for( j=0; j<3; j++ ) xn[off+j] = x[off+j] + k*v[off+j];
      v3_mul( v+off+3, k, aa );
      m3_from_aa( aa, Rk ); m3_from_aa( x+off+3, R0 );
      m3_mul( Rk, R0, Rn );
      m3_to_aa( Rn, xn+off+3 );
    } break;
    case RKFD_JOINT_SPHER: {
      double aa[3], Rk[9], R0[9], Rn[9];
      v3_mul( v+off, k, aa );
      m3_from_aa( aa, Rk ); m3_from_aa( x+off, R0 );
      m3_mul( Rk, R0, Rn );
      m3_to_aa( Rn, xn+off );
    } break;
    case RKFD_JOINT_FIXED: break;
    default: xn[off] = x[off] + k*v[off];
    }
  }
}

/* rkFDUpdate (reference src/rkfd_sim.c:560-566): zODE2Update with the Runge-Kutta-Gill scheme
 * [ZM zODE2 "Regular" + RKG, restated; UNVERIFIED-DEP], then the committing evaluation. */
int rkfdOracleUpdate(rkfdOracle *o)
{
  const double h = o->m->dt;
  const double s2 = sqrt( 2.0 );
  const double c21 = ( s2 - 1.0 )/2.0, c22 = ( 2.0 - s2 )/2.0;
  const double c31 = -s2/2.0, c32 = 1.0 + s2/2.0;
  const double w2 = 2.0 - s2, w3 = 2.0 + s2;
  int n = o->n, i, r = 0;
  double t = o->t;

  /* stage 1 */
  memcpy( o->k_v[0], o->vel, sizeof(double)*n );
  r |= evaluate( o, o->dis, o->vel, o->k_a[0], 0 );
  /* stage 2: x + h/2 k1 */
  cat_dis( o, o->dis, 0.5*h, o->k_v[0], o->xd );
  for( i=0; i<n; i++ ) o->xv[i] = o->vel[i] + 0.5*h*o->k_a[0][i];
  memcpy( o->k_v[1], o->xv, sizeof(double)*n );
  r |= evaluate( o, o->xd, o->xv, o->k_a[1], 0 );
  /* stage 3: x + h( c21 k1 + c22 k2 ) */
  for( i=0; i<n; i++ ){ o->tv[i] = c21*o->k_v[0][i] + c22*o->k_v[1][i]; o->ta[i] = c21*o->k_a[0][i] + c22*o->k_a[1][i]; }
  cat_dis( o, o->dis, h, o->tv, o->xd );
  for( i=0; i<n; i++ ) o->xv[i] = o->vel[i] + h*o->ta[i];
  memcpy( o->k_v[2], o->xv, sizeof(double)*n );
  r |= evaluate( o, o->xd, o->xv, o->k_a[2], 0 );
  /* stage 4: x + h( c31 k2 + c32 k3 ) */
  for( i=0; i<n; i++ ){ o->tv[i] = c31*o->k_v[1][i] + c32*o->k_v[2][i]; o->ta[i] = c31*o->k_a[1][i] + c32*o->k_a[2][i]; }
  cat_dis( o, o->dis, h, o->tv, o->xd );
  for( i=0; i<n; i++ ) o->xv[i] = o->vel[i] + h*o->ta[i];
  memcpy( o->k_v[3], o->xv, sizeof(double)*n );
  r |= evaluate( o, o->xd, o->xv, o->k_a[3], 0 );
  /* x += h/6 ( k1 + (2-sqrt2) k2 + (2+sqrt2) k3 + k4 ) */
  for( i=0; i<n; i++ ){
    o->tv[i] = o->k_v[0][i] + w2*o->k_v[1][i] + w3*o->k_v[2][i] + o->k_v[3][i];
    o->ta[i] = o->k_a[0][i] + w2*o->k_a[1][i] + w3*o->k_a[2][i] + o->k_a[3][i];
  }
  cat_dis( o, o->dis, h/6.0, o->tv, o->xd );
  memcpy( o->dis, o->xd, sizeof(double)*n );
  for( i=0; i<n; i++ ) o->vel[i] += h/6.0*o->ta[i];
  o->t = t + h;
  r |= eval_ref( o );
  return r;
}

/* nsteps x rkFDUpdate (the driver loop of reference example/chain/boxdrop_test.c:49-54) */
int rkfdOracleUpdateN(rkfdOracle *o, int nsteps)
{
  int r = 0, k;
  for( k=0; k<nsteps; k++ ) r |= rkfdOracleUpdate( o );
  return r;
}

/* number of KKT solves of the last Vert QP (diagnostic) */
int rkfdOracleLastQPIter(const rkfdOracle *o){ return o->last_qp_iter; }
/* Volume plugin, pair k of the last evaluation -> out[0..]: model pair, number of triangles, number of contact-plane
 * conditions, type; volume; center 3; norm 3; wrench 6; q 36; c 6; then (v 3, n 3) per condition, up to cap doubles */
int rkfdOracleGetVolumePair(const rkfdOracle *o, int k, double *out, int cap)
{
  const VolPair *vp;
  int i, n = 0;
  if( !o->vol_ready || k < 0 || k >= o->nvp ) return o->vol_ready ? -o->nvp - 1 : -1;
  vp = &o->vp[k];
#define PUT(x) do{ if( n < cap ) out[n] = (x); n++; }while(0)
  PUT( vp->pair ); PUT( vp->ntri ); PUT( vp->ncp ); PUT( o->vp_type[vp->pair] ); PUT( vp->volume );
  for( i=0; i<3; i++ ) PUT( vp->center[i] );
  for( i=0; i<3; i++ ) PUT( vp->norm[i] );
  for( i=0; i<6; i++ ) PUT( vp->wrench[i] );
  for( i=0; i<36; i++ ) PUT( vp->q[i] );
  for( i=0; i<6; i++ ) PUT( vp->c[i] );
  for( k=0; k<vp->ncp; k++ ){ for( i=0; i<3; i++ ) PUT( vp->cp[k].v[i] ); for( i=0; i<3; i++ ) PUT( vp->cp[k].n[i] ); }
#undef PUT
  return n;
}
int rkfdOracleVolumePairs(const rkfdOracle *o){ return o->vol_ready ? o->nvp : 0; }
int rkfdOracleVolumeLP(int mr, int n, const double *A, const double *b, const double *c, double *x){ return vol_lp( mr, n, A, b, c, x ); }
int rkfdOracleQPCycleStops(const rkfdOracle *o){ return o->qp_cycle_stops; }
int rkfdOracleVolumeGuardHits(const rkfdOracle *o){ return o->vol_guard_hits; }

/* test access to the two numerical building blocks of the Vert rigid branch */
void rkfdOraclePinvSolve(int n, const double *K, const double *rhs, double *x)
{
  double *k = (double *)malloc( sizeof(double)*n*n );
  memcpy( k, K, sizeof(double)*n*n );
  sym_pinv_solve( n, k, rhs, x );
  free( k );
}
int rkfdOracleQPASM(int n, int mc, int P, const double *q, const double *c, const double *nf, const double *d, double *ans, int *idx)
{
  return abs( qp_asm( n, mc, P, q, c, nf, d, ans, idx ) );
}

/* the last Vert QP (sizes with NULL pointers; then the data): min x'qx/2 + c'x s.t. nf x >= 0, its result and active set */
void rkfdOracleGetLastQP(const rkfdOracle *o, int *n, int *mc, double *q, double *c, double *nf, double *ans, int *idx)
{
  *n = o->qp_n; *mc = o->qp_mc;
  if( !q || !o->qp_q ) return;
  memcpy( q, o->qp_q, sizeof(double)*o->qp_n*o->qp_n ); memcpy( c, o->qp_c, sizeof(double)*o->qp_n );
  memcpy( nf, o->qp_nf, sizeof(double)*o->qp_mc*o->qp_n ); memcpy( ans, o->qp_ans, sizeof(double)*o->qp_n );
  memcpy( idx, o->qp_idx, sizeof(int)*o->qp_mc );
}
