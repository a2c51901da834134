/* rkfd_device.h - the batched rkFDUpdate step, one world instance per 64-lane wavefront.
 *
 * This is the MI355X-native restatement of the reference hot path
 *   rkFDUpdate -> zODE2Update(RKG) -> _rkFDUpdate -> {FK, CD, contact solver, ABA}
 *   (reference src/rkfd_sim.c:525-566; src/rkfd_util.c; src/rkfd_penalty.c:11-31;
 *    src/rkfd_mlcp.c), designed for a wavefront rather than translated:
 *   - all spatial quantities are expressed in ONE world frame (Pluecker coordinates at
 *     the world origin, (angular, linear) ordering), so the three ABA sweeps need no
 *     6x6 congruence transforms: parents simply sum their children's articulated
 *     inertias;
 *   - forward kinematics and link velocities are log-depth pointer-jumping scans over
 *     lanes (lane = link) instead of serial recursions;
 *   - sweep 2 / sweep 3 run level-synchronously, 8 lanes per link (lane = row of the
 *     6x6), up to 8 links of a level at once, 6x6 data staged in LDS, 6-lane
 *     reductions done with DPP;
 *   - contact candidates, penalty forces, MLCP probe columns and the PGS residual
 *     update are lane-parallel (lane = candidate / contact / probe column / row).
 * The file compiles for gfx950 (hipcc) and, with -DRKFD_EMU, under a 64-thread lane
 * emulator that exists only so that tests can exercise the kernel logic without a GPU
 * (tests/emu; never part of the product library).
 */
#ifndef RKFD_DEVICE_H
#define RKFD_DEVICE_H

#include <math.h>
#include "rkfd_model.h"
#include "rkfd_devmodel.h"

#ifdef RKFD_EMU
#  define RKFD_DEV static inline
   int    rkfd_emu_lane(void);
   void   rkfd_emu_sync(void);
   double rkfd_emu_g8sum(double x);
   double rkfd_emu_bcast(double x, int src);
   unsigned long long rkfd_emu_ballot(int pred);
#  define LANE()        rkfd_emu_lane()
#  define SYNC()        rkfd_emu_sync()
   double rkfd_emu_g8bcast(double x, int k);
#  define G8SUM(x)      rkfd_emu_g8sum(x)
#  define G8SUM2(x,y)   do{ (x) = rkfd_emu_g8sum(x); (y) = rkfd_emu_g8sum(y); }while(0)
#  define G8BCAST(x,k)  rkfd_emu_g8bcast(x,k)
#  define RKFD_RCP(x)   ( 1.0/(x) )
#  define LDS_FENCE()   rkfd_emu_sync()
#  define BCAST(x,l)    rkfd_emu_bcast(x,l)
#  define BALLOT(p)     rkfd_emu_ballot(p)
#else
#  define RKFD_DEV __device__ __forceinline__
#  define LANE()        ((int)threadIdx.x)
/* One workgroup is one wavefront: lanes exchange data through LDS in program order, so a
 * "barrier" only has to (a) stop the compiler from moving LDS accesses across it and (b) wait
 * for the wave's own outstanding LDS operations.  __syncthreads() would also drain vmcnt (the
 * schedule-record prefetches), which is exactly the latency the prefetch is meant to hide. */
#  define SYNC()        asm volatile( "s_waitcnt lgkmcnt(0)" ::: "memory" )
RKFD_DEV double rkfd_dpp_xor1(double x)
{
  int lo = __double2loint( x ), hi = __double2hiint( x );
  lo = __builtin_amdgcn_update_dpp( lo, lo, 0xB1, 0xF, 0xF, false ); /* quad_perm [1,0,3,2] */
  hi = __builtin_amdgcn_update_dpp( hi, hi, 0xB1, 0xF, 0xF, false );
  return __hiloint2double( hi, lo );
}
RKFD_DEV double rkfd_dpp_xor2(double x)
{
  int lo = __double2loint( x ), hi = __double2hiint( x );
  lo = __builtin_amdgcn_update_dpp( lo, lo, 0x4E, 0xF, 0xF, false ); /* quad_perm [2,3,0,1] */
  hi = __builtin_amdgcn_update_dpp( hi, hi, 0x4E, 0xF, 0xF, false );
  return __hiloint2double( hi, lo );
}
RKFD_DEV double rkfd_dpp_hmirror(double x)
{
  int lo = __double2loint( x ), hi = __double2hiint( x );
  lo = __builtin_amdgcn_update_dpp( lo, lo, 0x141, 0xF, 0xF, false ); /* row_half_mirror */
  hi = __builtin_amdgcn_update_dpp( hi, hi, 0x141, 0xF, 0xF, false );
  return __hiloint2double( hi, lo );
}
/* sum over the aligned group of 8 lanes, result in every lane of the group */
RKFD_DEV double rkfd_g8sum(double x)
{
  x += rkfd_dpp_xor1( x );
  x += rkfd_dpp_xor2( x );
  x += rkfd_dpp_hmirror( x );
  return x;
}
/* two independent 8-lane sums, interleaved so that their DPP chains overlap */
RKFD_DEV void rkfd_g8sum2(double &x, double &y)
{
  double a = rkfd_dpp_xor1( x ), b = rkfd_dpp_xor1( y );
  x += a; y += b;
  a = rkfd_dpp_xor2( x ); b = rkfd_dpp_xor2( y );
  x += a; y += b;
  a = rkfd_dpp_hmirror( x ); b = rkfd_dpp_hmirror( y );
  x += a; y += b;
}
/* broadcast lane src (wave-uniform) to every lane */
RKFD_DEV double rkfd_bcast(double x, int src)
{
  int lo = __builtin_amdgcn_readlane( __double2loint( x ), src );
  int hi = __builtin_amdgcn_readlane( __double2hiint( x ), src );
  return __hiloint2double( hi, lo );
}
/* broadcast lane k (compile-time 0..7) of every aligned 8-lane group to the whole group:
 * ds_swizzle in bit mode, lane' = ( lane & 0x18 ) | k within each half-wave; no LDS storage */
template<int K> RKFD_DEV double rkfd_g8bcast(double x)
{
  int lo = __builtin_amdgcn_ds_swizzle( __double2loint( x ), ( K << 5 ) | 0x18 );
  int hi = __builtin_amdgcn_ds_swizzle( __double2hiint( x ), ( K << 5 ) | 0x18 );
  return __hiloint2double( hi, lo );
}
/* reciprocal: v_rcp_f64 + two Newton steps (relative error ~1e-16) instead of the IEEE division sequence */
RKFD_DEV double rkfd_rcp(double x)
{
  double r = __builtin_amdgcn_rcp( x );
  r = fma( r, fma( -x, r, 1.0 ), r );
  r = fma( r, fma( -x, r, 1.0 ), r );
  return r;
}
#  define G8SUM(x)      rkfd_g8sum(x)
#  define G8SUM2(x,y)   rkfd_g8sum2(x,y)
#  define G8BCAST(x,k)  rkfd_g8bcast<k>(x)
#  define RKFD_RCP(x)   rkfd_rcp(x)
/* compiler-only fence: LDS operations of one wavefront execute in program order */
#  define LDS_FENCE()   asm volatile( "" ::: "memory" )
#  define BCAST(x,l)    rkfd_bcast(x,l)
#  define BALLOT(p)     __ballot(p)
#endif

#define RKFD_DEV_TOL RKFD_TOL

/* RELOAD(p): makes the compiler forget what it knows about pointer p.  The per-lane model constants
 * (link frames, inertias, motor data ...) are the same in every evaluation, so LLVM hoists their
 * loads out of the step loop and then has to SPILL ~35 doubles per lane to scratch - HBM write
 * traffic an order of magnitude above the algorithmic bytes.  Re-reading them from L2 is cheaper. */
#ifdef RKFD_EMU
#  define RELOAD(p) (p)
#else
template<class T> RKFD_DEV const T *rkfd_reload(const T *p){ asm volatile( "" : "+s"(p) ); return p; }
#  define RELOAD(p) rkfd_reload(p)
#endif

/* optional in-kernel phase timing (diagnostic launches only: rkfdBatchProfile) */
#define RKFD_NPROF 24
#ifdef RKFD_EMU
#  define RKFD_CLOCK() 0ull
#else
#  define RKFD_CLOCK() ( (unsigned long long)__builtin_amdgcn_s_memtime() )
#endif

/* ------------------------------------------------------------------------ */
/* 3-vector helpers on plain arrays */
RKFD_DEV void d_cross(const double *a, const double *b, double *c)
{
  double x = a[1]*b[2]-a[2]*b[1], y = a[2]*b[0]-a[0]*b[2], z = a[0]*b[1]-a[1]*b[0];
  c[0]=x; c[1]=y; c[2]=z;
}
RKFD_DEV double d_dot(const double *a, const double *b){ return a[0]*b[0]+a[1]*b[1]+a[2]*b[2]; }
RKFD_DEV void d_mulv(const double *m, const double *v, double *r)
{
  double x = m[0]*v[0]+m[1]*v[1]+m[2]*v[2], y = m[3]*v[0]+m[4]*v[1]+m[5]*v[2], z = m[6]*v[0]+m[7]*v[1]+m[8]*v[2];
  r[0]=x; r[1]=y; r[2]=z;
}
RKFD_DEV void d_tmulv(const double *m, const double *v, double *r)
{
  double x = m[0]*v[0]+m[3]*v[1]+m[6]*v[2], y = m[1]*v[0]+m[4]*v[1]+m[7]*v[2], z = m[2]*v[0]+m[5]*v[1]+m[8]*v[2];
  r[0]=x; r[1]=y; r[2]=z;
}
RKFD_DEV void d_mul33(const double *a, const double *b, double *c)
{
  double t[9];
#pragma unroll
  for( int i=0; i<3; i++ )
#pragma unroll
    for( int j=0; j<3; j++ )
      t[3*i+j] = a[3*i]*b[j] + a[3*i+1]*b[3+j] + a[3*i+2]*b[6+j];
#pragma unroll
  for( int i=0; i<9; i++ ) c[i] = t[i];
}
/* ------------------------------------------------------------------------ */
/* compact sin/cos and atan2 for joint-angle sized arguments.  The library versions inline a
 * Payne-Hanek slow path (v_trig_preop) that costs registers and code for arguments a robot never
 * has; these use a two-term Cody-Waite reduction by pi/2 and the classic fdlibm kernel
 * polynomials (|error| < 1 ulp for |x| < 1e5), and an fdlibm-style atan. */
/* the polynomial coefficients live in constant memory and are fetched with scalar loads when a
 * function runs: as 64-bit literals the compiler materialises them in VGPR pairs, hoists them out
 * of the step loop and then spills them to scratch */
#ifdef RKFD_EMU
static const double rkfd_kc[] = {
#else
__constant__ double rkfd_kc[] = {
#endif
  /*  0 */ 6.36619772367581382433e-01, 1.57079632673412561417e+00, 6.07710050650619224932e-11,
  /*  3 sin */ -1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04,
               2.75573137070700676789e-06, -2.50507602534068634195e-08, 1.58969099521155010221e-10,
  /*  9 cos */ 4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05,
               -2.75573143513906633035e-07, 2.08757232129817482790e-09, -1.13596475577881948265e-11,
  /* 15 atan hi/lo */ 4.63647609000806093515e-01, 2.26987774529616870924e-17, 7.85398163397448278999e-01, 3.06161699786838301793e-17,
               9.82793723247329054082e-01, 1.39033110312309984516e-17, 1.57079632679489655800e+00, 6.12323399573676603587e-17,
  /* 23 atan odd */ 3.33333333333329318027e-01, 1.42857142725034663711e-01, 9.09088713343650656196e-02,
               6.66107313738753120669e-02, 4.97687799461593236017e-02, 1.62858201153657823623e-02,
  /* 29 atan even */ -1.99999999998764832476e-01, -1.11111104054623557880e-01, -7.69187620504482999495e-02,
               -5.83357013379057348645e-02, -3.65315727442169155270e-02,
  /* 34 */ 3.14159265358979311600e+00
};
RKFD_DEV void d_sincos(double x, double *sn, double *cs)
{
  const double *K = RELOAD( (const double *)rkfd_kc );
  const double k = rint( x*K[0] );
  double r = fma( -k, K[1], x );
  r = fma( -k, K[2], r );
  const double z = r*r;
  const double ps = K[3] + z*( K[4] + z*( K[5] + z*( K[6] + z*( K[7] + z*K[8] ) ) ) );
  const double pc = K[9] + z*( K[10] + z*( K[11] + z*( K[12] + z*( K[13] + z*K[14] ) ) ) );
  const double s0 = fma( r*z, ps, r );
  const double c0 = fma( z*z, pc, fma( -0.5, z, 1.0 ) );
  const int q = (int)k & 3;
  const double s1 = ( q & 1 ) ? c0 : s0, c1 = ( q & 1 ) ? s0 : c0;
  *sn = ( q & 2 ) ? -s1 : s1;
  *cs = ( ( q + 1 ) & 2 ) ? -c1 : c1;
}
RKFD_DEV double d_atan_pos(double x)   /* x >= 0 */
{
  /* fdlibm atan: reduce to |t| <= 7/16 around 0, 0.5, 1, 1.5, inf */
  const double *K = RELOAD( (const double *)rkfd_kc );
  double hi, lo, t;
  if( x < 0.4375 ){ hi = 0; lo = 0; t = x; }
  else if( x < 0.6875 ){ hi = K[15]; lo = K[16]; t = ( 2.0*x - 1.0 )/( 2.0 + x ); }
  else if( x < 1.1875 ){ hi = K[17]; lo = K[18]; t = ( x - 1.0 )/( x + 1.0 ); }
  else if( x < 2.4375 ){ hi = K[19]; lo = K[20]; t = ( x - 1.5 )/( 1.0 + 1.5*x ); }
  else { hi = K[21]; lo = K[22]; t = -1.0/x; }
  const double z = t*t, w = z*z;
  const double s1 = z*( K[23] + w*( K[24] + w*( K[25] + w*( K[26] + w*( K[27] + w*K[28] ) ) ) ) );
  const double s2 = w*( K[29] + w*( K[30] + w*( K[31] + w*( K[32] + w*K[33] ) ) ) );
  return hi - ( ( t*( s1 + s2 ) - lo ) - t );
}
RKFD_DEV double d_atan2_ypos(double y, double x)   /* y >= 0 */
{
  const double *K = RELOAD( (const double *)rkfd_kc );
  if( x > 0 ) return d_atan_pos( y/x );
  if( x < 0 ) return K[34] - d_atan_pos( y/( -x ) );
  return y > 0 ? K[21] : 0.0;
}

RKFD_DEV void d_from_aa(const double *aa, double *m)
{
  double th = sqrt( d_dot( aa, aa ) );
  if( th < RKFD_DEV_TOL ){
    m[0]=1; m[1]=0; m[2]=0; m[3]=0; m[4]=1; m[5]=0; m[6]=0; m[7]=0; m[8]=1;
    return;
  }
  double s, c;
  d_sincos( th, &s, &c );
  const double k = 1-c, ith = 1.0/th;
  double x = aa[0]*ith, y = aa[1]*ith, z = aa[2]*ith;
  m[0] = c+k*x*x;   m[1] = k*x*y-s*z; m[2] = k*x*z+s*y;
  m[3] = k*x*y+s*z; m[4] = c+k*y*y;   m[5] = k*y*z-s*x;
  m[6] = k*x*z-s*y; m[7] = k*y*z+s*x; m[8] = c+k*z*z;
}
RKFD_DEV void d_to_aa(const double *m, double *aa)
{
  double l[3] = { m[7]-m[5], m[2]-m[6], m[3]-m[1] };
  double a = sqrt( d_dot( l, l ) );
  double th = d_atan2_ypos( a, m[0]+m[4]+m[8]-1.0 );
  if( a < RKFD_DEV_TOL ){ aa[0]=aa[1]=aa[2]=0; return; }
  double k = th/a;
  aa[0] = l[0]*k; aa[1] = l[1]*k; aa[2] = l[2]*k;
}
RKFD_DEV void d_ortho_space(const double *n, double *t1, double *t2)
{
  int k = 0;
  if( fabs(n[1]) < fabs(n[k]) ) k = 1;
  if( fabs(n[2]) < fabs(n[k]) ) k = 2;
  double e[3] = { k==0 ? 1.0 : 0.0, k==1 ? 1.0 : 0.0, k==2 ? 1.0 : 0.0 };
  double d = d_dot( e, n );
  t1[0] = e[0]-d*n[0]; t1[1] = e[1]-d*n[1]; t1[2] = e[2]-d*n[2];
  double l = sqrt( d_dot( t1, t1 ) );
  t1[0] /= l; t1[1] /= l; t1[2] /= l;
  d_cross( n, t1, t2 );
}
/* spatial motion cross product v x m and force cross product v x* f, (ang, lin) ordering */
RKFD_DEV void d_crm(const double *v, const double *m, double *r)
{
  double a[3], b[3], c[3];
  d_cross( v, m, a ); d_cross( v, m+3, b ); d_cross( v+3, m, c );
  r[0]=a[0]; r[1]=a[1]; r[2]=a[2]; r[3]=b[0]+c[0]; r[4]=b[1]+c[1]; r[5]=b[2]+c[2];
}
RKFD_DEV void d_crf(const double *v, const double *f, double *r)
{
  double a[3], b[3], c[3];
  d_cross( v, f, a ); d_cross( v+3, f+3, b ); d_cross( v, f+3, c );
  r[0]=a[0]+b[0]; r[1]=a[1]+b[1]; r[2]=a[2]+b[2]; r[3]=c[0]; r[4]=c[1]; r[5]=c[2];
}

/* ------------------------------------------------------------------------ */
/* motor model (see oracle/rkfd_oracle.c for the RoKi call sites it restates) */
RKFD_DEV double d_clamp(double x, double lo, double hi){ return x < lo ? lo : ( x > hi ? hi : x ); }

/* ------------------------------------------------------------------------ */
/* LDS carve-up for one instance */
typedef struct {
  double *q, *qd, *acc;           /* [ndof] each                                         */
  double *tmp;                    /* [ndof] scratch of rkfd_cat_dis: ALIASES V (dead between evaluations) */
  double *S;                      /* [NL*6]  joint axis (ang, lin)                        */
  double *V;                      /* [NL*6]  spatial velocity (kinematics .. rkfd_phase_bvel)               */
  double *U;                      /* [NL*6]  Ia S, written by sweep 2: ALIASES V                            */
  double *PB;                     /* [NL*6]  own bias force minus the external wrenches (kinematics .. sweep 2) */
  double *AC;                     /* [NL*6]  spatial acceleration, written by sweep 3: ALIASES PB           */
  double *C;                      /* [NL*6]  velocity-product acceleration (kinematics .. sweep 3)          */
  double *PA;                     /* [NL*6]  bias force handed to the parent (sweep 2)                      */
  double *XA, *XB;                /* [NL*6] each: world frames, R rows 0-1 | R row 2, p.  Valid from the kinematics
                                     phase to the end of the collision phase: XA ALIASES PA, XB the Ia pool */
  double *MS;                     /* [NL*4]  Dinv, u, tau, jm                             */
  double *IST;                    /* [NL*14] inertia staging: A = Iw + m(|r|^2 1 - r r') (xx,xy,xz,yy,yz,zz),
                                     +m r (3), -m r (3), m, 0: every entry of the 6x6 is one of these */
  double *POOL;                   /* [npool*36] Ia of links whose parent gathers through LDS */
  double *CHOL;                   /* [nfloat*36] articulated inertia / Cholesky factor of float joints */
  double *XF;                     /* [nfloat*12] float joints: world orientation of the joint-origin frame (9), link position (3) */
  double *CX, *AX, *RW, *PRO;     /* per ACTIVE contact slot (capacity maxact): 3, 9, 3, 3 */
  double *REF;                    /* stick anchors (state): per active slot              */
  double *RTMP;                   /* [maxact*3] copy of REF while the slots are re-assigned; only when ncand > 64 */
  double *CF;                     /* contact forces (output): per active slot              */
  double *MA, *MB, *MF, *PU;      /* MLCP: [M*(M+1)] (ALIASES IST|POOL), [M], [M], [nside*npurow*M] (ALIASES C|PA when it fits) */
  int *act, *typ, *lrg, *lel, *tgt, *cnt;
  int *asl;                       /* [NC] active-contact slot of a candidate              */
  int *LI;                        /* [NL] packed link info (RKFD_LI_*)                    */
  int *CIp, *CFO;                 /* [NC] packed candidate info, first plane              */
  int *CHI;                       /* [NL] children lists (CSR values; offsets in the schedule)     */
  int *PSL;                       /* [NL] pool slot of a link (-1 none)                   */
  unsigned char *PL;              /* [NL*nlevel] ancestor at depth d (MLCP only), one byte each */
} rkfdLds;

RKFD_DEV void rkfd_lds_carve(rkfdLds *L, void *base, int NL, int ND, int NC, int M, int nlevel, int npool, int nfloat, int maxact, int nside, int pu_alias, int npurow)
/* must match the byte count computed in rkfd_devmodel.cpp */
{
  double *d = (double *)base;
  L->q = d; d += ND; L->qd = d; d += ND; L->acc = d; d += ND;
  L->S = d; d += NL*6;
  L->V = d; L->U = d; L->tmp = d; d += NL*6;
  L->PB = d; L->AC = d; d += NL*6;
  L->C = d; d += NL*6;
  L->PA = d; L->XA = d; d += NL*6;
  L->MS = d; d += NL*4;
  {
    const int pool = 36*npool > 6*NL ? 36*npool : 6*NL;
    int stage = 14*NL + pool;
    L->IST = d; L->POOL = d + 14*NL; L->XB = d + 14*NL; L->MA = d;
    if( M*(M+1) > stage ) stage = M*(M+1);
    d += stage;
  }
  L->CHOL = d; d += 36*nfloat; L->XF = d; d += 12*nfloat;
  L->CX = d; d += maxact*3; L->AX = d; d += maxact*9; L->RW = d; d += maxact*3; L->PRO = d; d += maxact*3;
  L->REF = d; d += maxact*3; L->RTMP = d; if( NC > RKFD_WAVE ) d += maxact*3;
  L->CF = d; d += maxact*3;
  L->MB = d; d += M; L->MF = d; d += M;
  /* probe scratch: lives while the contact problem is set up and solved, when C and PA are dead */
  if( pu_alias ) L->PU = L->C; else { L->PU = d; d += nside*npurow*M; }
  int *ip = (int *)d;
  L->act = ip; ip += NC; L->typ = ip; ip += NC; L->asl = ip; ip += NC; L->CIp = ip; ip += NC; L->CFO = ip; ip += NC;
  L->lrg = ip; ip += maxact; L->lel = ip; ip += maxact; L->tgt = ip; ip += 2*maxact; L->cnt = ip; ip += 8;
  L->LI = ip; ip += NL; L->CHI = ip; ip += NL; L->PSL = ip; ip += NL;
  L->PL = (unsigned char *)ip;
}

/* per-lane state that only lane = link ever touches: kept in registers for the whole launch */
typedef struct { double min, pivp; int pivt; } rkfdLaneLink;

/* the stick anchors REF live per active-contact slot */
#define RIDX(j) ( L.asl[j] )

/* packed description of one moving side of a rigid contact (built per evaluation in L->tgt) */
#define RKFD_CS_LINK(e)   ( (int)( (e) & 0xFF ) )
#define RKFD_CS_DEPTH(e)  ( (int)( ( (e) >> 8 ) & 0x3F ) )
#define RKFD_CS_TOP(e)    ( (int)( ( (e) >> 14 ) & 0xFF ) )
#define RKFD_CS_D0(e)     ( (int)( ( (e) >> 22 ) & 0x7F ) )
#define RKFD_CS_FLOAT(e)  ( (int)( ( (e) >> 29 ) & 1 ) )
#define RKFD_CS_SIDE(e)   ( (int)( ( (e) >> 30 ) & 1 ) )
#define RKFD_CS_VALID(e)  ( (int)( (e) >> 31 ) )

/* counters in L->cnt */
#define CNT_NRG 0
#define CNT_NEL 1
#define CNT_NTGT 2
#define CNT_OVF 3

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
  double R[9], p[3], Rj[9], vJ[6], qd1 = 0, qdf[6] = {0,0,0,0,0,0};
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
      double q = L.q[off], s, c;
      d_sincos( q, &s, &c );
      double Rz[9] = { c,-s,0, s,c,0, 0,0,1 };
      d_mul33( o, Rz, R );
      qd1 = L.qd[off];
    } else if( jt == RKFD_JOINT_PRISM ){
      double q = L.q[off];
      p[0] += q*o[2]; p[1] += q*o[5]; p[2] += q*o[8];
      qd1 = L.qd[off];
    } else if( jt == RKFD_JOINT_FLOAT ){
      double qq[6], t[3];
#pragma unroll
      for( int k=0; k<6; k++ ){ qq[k] = L.q[off+k]; qdf[k] = L.qd[off+k]; }
      d_from_aa( qq+3, Rj );
      d_mul33( o, Rj, R );
      d_mulv( o, qq, t );
      p[0] += t[0]; p[1] += t[1]; p[2] += t[2];
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
          jm = RELOAD( m.mot_inertia )[i]*gear*gear;
          tin = admit*gk*d_clamp( in, RELOAD( m.mot_vmin )[i], RELOAD( m.mot_vmax )[i] );
          treg = admit*gk*gk*qd1;
          tf = jm*( -qd1/m.dt ) - tin + treg + ll.pivp;
          double fmax;
          if( ll.pivt == RKFD_SF ) fmax = RELOAD( m.sfric )[i];
          else {
            const double q = L.q[off];
            const double sg = qd1 > 0 ? 1.0 : ( qd1 < 0 ? -1.0 : 0.0 );
            fmax = -RELOAD( m.stiff )[i]*q - RELOAD( m.visc )[i]*qd1 - RELOAD( m.coulomb )[i]*sg;
          }
          fmax = fabs( fmax );
          int newt;
          if( fabs( tf ) > fmax ){ tf = tf > 0 ? fmax : -fmax; newt = RKFD_KF; }
          else newt = RKFD_SF;
          /* the pivot type is committed by the caller when doUpRef (stored in MS slot 1 as a flag) */
          L.MS[4*i+1] = (double)newt;
        } else if( mt == RKFD_MOTOR_TRQ ){
          tin = d_clamp( in, RELOAD( m.mot_vmin )[i], RELOAD( m.mot_vmax )[i] );
        }
        tau = tin - treg + tf;
        /* driving torque without the inertia term + friction, for rkFDUpdateJointPrevDrivingTrq */
        L.MS[4*i+0] = tin - treg + tf;
      }
      L.MS[4*i+2] = tau;
      L.MS[4*i+3] = jm;
    }
  }
  SYNC();
  KST(20);
#undef KST
}

/* ------------------------------------------------------------------------ */
/* in-place Cholesky of the 6x6 at A (row-major, lower part used), one lane.  The diagonal
 * stores 1/L_jj so that the factorisation and the solves multiply instead of dividing. */
RKFD_DEV void d_chol6_inplace(double *A)
{
  /* the lower triangle is pulled into registers in one batch of loads, factored there and written back */
  double a[6][6];
#pragma unroll
  for( int i=0; i<6; i++ )
#pragma unroll
    for( int k=0; k<6; k++ ) if( k <= i ) a[i][k] = A[6*i+k];
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
    for( int k=0; k<6; k++ ) if( k <= i ) A[6*i+k] = a[i][k];
}
/* forward substitution y = L^-1 b and back substitution x = L^-T y with that factor */
RKFD_DEV void d_chol6_fwd(const double *Lm, const double *b, double *y)
{
#pragma unroll
  for( int i=0; i<6; i++ ){
    double s = b[i];
#pragma unroll
    for( int k=0; k<6; k++ ) if( k < i ) s -= Lm[6*i+k]*y[k];
    y[i] = s*Lm[6*i+i];
  }
}
RKFD_DEV void d_chol6_back(const double *Lm, const double *y, double *x)
{
#pragma unroll
  for( int i=5; i>=0; i-- ){
    double s = y[i];
#pragma unroll
    for( int k=0; k<6; k++ ) if( k > i ) s -= Lm[6*k+i]*x[k];
    x[i] = s*Lm[6*i+i];
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
  const int *p = m.sched + ( (size_t)( t+2 )*8 + g )*4;
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
RKFD_DEV void rkfd_pre2_load(const rkfdLds &L, int i, int rr, const int *ro, rkfdPre2 &p)
{
#pragma unroll
  for( int k=0; k<6; k++ ){ p.S[k] = L.S[6*i+k]; p.c[k] = L.C[6*i+k]; }
  p.S_r = L.S[6*i+rr];
  p.pb = L.PB[6*i+rr];
  p.tau = L.MS[4*i+2]; p.jm = L.MS[4*i+3];
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
    /* operands of this iteration (with two waves per SIMD the other wave covers the LDS latency;
     * a second, prefetched operand set would cost ~60 VGPRs) */
    rkfdPre2 pre;
    rkfd_pre2_load( L, rec.i >= 0 ? rec.i : 0, rr, ro, pre );
    QST(8);
    const bool onl = rec.i >= 0;
    const bool on = onl && r < 6;
    const int i = onl ? rec.i : 0;
    const int jt = onl ? RKFD_LI_JT( rec.li ) : RKFD_JOINT_FIXED;
    const bool is1 = jt == RKFD_JOINT_REVOL || jt == RKFD_JOINT_PRISM;
    const bool isf = jt == RKFD_JOINT_FLOAT;
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
        const int ch = L.CHI[rec.coff+cc];
        pr += L.PA[6*ch+rr];
        const int ps = L.PSL[ch];
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
#pragma unroll
        for( int k=0; k<6; k++ ) L.POOL[36*REC_POOL( rec )+6*rr+k] = row[k];
      }
      if( isf ){
#pragma unroll
        for( int k=0; k<6; k++ ) L.CHOL[36*REC_FSLOT( rec )+6*rr+k] = row[k];
      }
      if( is1 ) L.U[6*i+rr] = U_r;
      L.PA[6*i+rr] = pa;
      if( isf ) L.U[6*i+rr] = pr;   /* float: the U slot keeps the bias pA */
      if( rr == 0 && is1 ){
        L.MS[4*i+0] = Dinv;
        L.MS[4*i+1] = u;
      }
    }
    LDS_FENCE();
    QST(12);
    if( isf && onl && r == 0 ) d_chol6_inplace( &L.CHOL[36*REC_FSLOT( rec )] );
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
  p.Dinv = L.MS[4*i+0]; p.u = L.MS[4*i+1];
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
    const int jt = onl ? RKFD_LI_JT( rec.li ) : RKFD_JOINT_FIXED;
    const bool is1 = jt == RKFD_JOINT_REVOL || jt == RKFD_JOINT_PRISM;
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
        if( delta ) L.acc[off] += qdd; else L.acc[off] = qdd;
      }
    } else if( jt == RKFD_JOINT_FLOAT ){
      if( onl && r == 0 ){
        /* a = IA^-1 ( -pA ); joint acceleration from a - a_parent - c */
        double rhs[6], x[6], d[6], Row[9], p[3];
        if( delta ){
#pragma unroll
          for( int k=0; k<6; k++ ) rhs[k] = L.U[6*i+k];
          d_chol6_back( &L.CHOL[36*REC_FSLOT( rec )], rhs, x );
        } else {
          double yv[6];
#pragma unroll
          for( int k=0; k<6; k++ ) rhs[k] = -L.U[6*i+k];
          d_chol6_fwd( &L.CHOL[36*REC_FSLOT( rec )], rhs, yv );
          d_chol6_back( &L.CHOL[36*REC_FSLOT( rec )], yv, x );
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
        if( delta ){
          L.acc[off] += o1[0]; L.acc[off+1] += o1[1]; L.acc[off+2] += o1[2];
          L.acc[off+3] += o2[0]; L.acc[off+4] += o2[1]; L.acc[off+5] += o2[2];
        } else {
          L.acc[off] = o1[0]; L.acc[off+1] = o1[1]; L.acc[off+2] = o1[2];
          L.acc[off+3] = o2[0]; L.acc[off+4] = o2[1]; L.acc[off+5] = o2[2];
        }
      }
    }
    LDS_FENCE();
    if( jt == RKFD_JOINT_FLOAT ) a = L.AC[6*i+rr];
    if( on && jt != RKFD_JOINT_FLOAT ) L.AC[6*i+rr] = a;
    ca = a;
  }
  SYNC();
}

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
  if( m.ncand > RKFD_WAVE ){
    for( int k=lane; k<3*m.maxact; k+=RKFD_WAVE ) L.RTMP[k] = L.REF[k];
    oldref = L.RTMP;
    SYNC();
  }
  /* candidates are swept 64 at a time; slots and list positions keep candidate order */
  for( int c0=0; c0<m.ncand; c0+=RKFD_WAVE ){
    const bool on = c0+lane < m.ncand;
    const int j = on ? c0+lane : 0;
    int is_act = 0, is_rg = 0, is_el = 0, fbest = -1;
    double x[3] = {0,0,0}, y[3] = {0,0,0}, smax = -HUGE_VAL, RB[9], pB[3];
    const int cinf = L.CIp[j];
#pragma unroll
    for( int k=0; k<9; k++ ) RB[k] = 0;
    pB[0] = pB[1] = pB[2] = 0;
    if( on ){
      const int la = RKFD_CI_A( cinf ), lb = RKFD_CI_B( cinf );
      double RA[9], pA[3], vl[3], rr[3];
#pragma unroll
      for( int k=0; k<6; k++ ){ RA[k] = L.XA[6*la+k]; RB[k] = L.XA[6*lb+k]; }
#pragma unroll
      for( int k=0; k<3; k++ ){ RA[6+k] = L.XB[6*la+k]; RB[6+k] = L.XB[6*lb+k]; }
#pragma unroll
      for( int k=0; k<3; k++ ){ pA[k] = L.XB[6*la+3+k]; pB[k] = L.XB[6*lb+3+k]; vl[k] = RELOAD( m.cand_vert )[3*j+k]; }
      d_mulv( RA, vl, x );
      x[0] += pA[0]; x[1] += pA[1]; x[2] += pA[2];
      rr[0] = x[0]-pB[0]; rr[1] = x[1]-pB[1]; rr[2] = x[2]-pB[2];
      d_tmulv( RB, rr, y );
      const int f0 = L.CFO[j], nf = RKFD_CI_NF( cinf );
      for( int f=f0; f<f0+nf; f++ ){
        const double sd = m.planes[4*f]*y[0] + m.planes[4*f+1]*y[1] + m.planes[4*f+2]*y[2] - m.planes[4*f+3];
        if( sd > smax ){ smax = sd; fbest = f; }
      }
      is_act = fbest >= 0 && smax < RKFD_DEV_TOL;
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
        L.CF[3*slot] = 0; L.CF[3*slot+1] = 0; L.CF[3*slot+2] = 0;
        d_mulv( RB, ref, rw );
        L.RW[3*slot] = rw[0]+pB[0]; L.RW[3*slot+1] = rw[1]+pB[1]; L.RW[3*slot+2] = rw[2]+pB[2];
        d_ortho_space( nw, t1, t2 );
#pragma unroll
        for( int k=0; k<3; k++ ){ L.AX[9*slot+k] = nw[k]; L.AX[9*slot+3+k] = t1[k]; L.AX[9*slot+6+k] = t2[k]; }
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
RKFD_DEV void rkfd_push_wrenches(const rkfdDevModel &m, const rkfdLds &L, const int *list, int n)
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
  const double *ax = &L.AX[9*L.asl[j]];
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
    if( doUpRef ) L.typ[j] = RKFD_SF;
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
      vr[k] = va[k]-vb[k];
      f[k] = -E*( x[k]-L.RW[3*L.asl[j]+k] ) + kv*vr[k];
    }
    if( d_dot( f, &L.AX[9*L.asl[j]] ) < 0.0 ){
      f[0] = f[1] = f[2] = 0;
    } else {
      d_modify_friction( m, L, j, vr, f, doUpRef );
    }
    { const int sl_ = L.asl[j]; L.CF[3*sl_] = f[0]; L.CF[3*sl_+1] = f[1]; L.CF[3*sl_+2] = f[2]; }
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
#pragma unroll
    for( int i=0; i<3; i++ ){
      const double *ax = &L.AX[9*sl+3*i];
      bv[i]   = ax[0]*( va[0]-vb[0] ) + ax[1]*( va[1]-vb[1] ) + ax[2]*( va[2]-vb[2] );
      bv[3+i] = ax[0]*( ca[0]-cb[0] ) + ax[1]*( ca[1]-cb[1] ) + ax[2]*( ca[2]-cb[2] );
    }
  }
  SYNC();
}

/* projected Gauss-Seidel for up to RKFD_PGS_NC contacts with the lane's three matrix rows held in
 * registers (3 x 3*RKFD_PGS_NC doubles): the contact loop is unrolled, so the rows are indexed
 * statically, the broadcasts read fixed lanes and the dependent path of an update is ALU only.
 * Same arithmetic and update order as the general loop in rkfd_phase_mlcp. */
#define RKFD_PGS_NC 4
RKFD_DEV void rkfd_pgs_registers(const double *Arow, int ld, int nc, int max_iter, bool on, int lane, double mu,
                                 double in_, double i1, double i2, double &rn, double &r1, double &r2, double &fn, double &f1, double &f2)
{
  double A0[3*RKFD_PGS_NC], A1[3*RKFD_PGS_NC], A2[3*RKFD_PGS_NC];
#pragma unroll
  for( int k=0; k<3*RKFD_PGS_NC; k++ ){
    const bool in = on && k < 3*nc;
    A0[k] = in ? Arow[k] : 0.0; A1[k] = in ? Arow[ld+k] : 0.0; A2[k] = in ? Arow[2*ld+k] : 0.0;
  }
  for( int it=0; it<max_iter; it++ ){
#pragma unroll
    for( int c=0; c<RKFD_PGS_NC; c++ ){
      if( c < nc ){
        double ff = fn - rn*in_;
        if( ff < RKFD_DEV_TOL ) ff = 0.0;
        const double dl = BCAST( ff - fn, c );
        if( lane == c ) fn = ff;
        rn = fma( A0[3*c], dl, rn ); r1 = fma( A1[3*c], dl, r1 ); r2 = fma( A2[3*c], dl, r2 );
      }
    }
#pragma unroll
    for( int c=0; c<RKFD_PGS_NC; c++ ){
      if( c < nc ){
        const double ff0 = f1 - r1*i1, ff1 = f2 - r2*i2;
        const double fnorm = ff0*ff0 + ff1*ff1;
        double fs = mu*fn; fs = fs*fs;
        /* only the decision of lane c matters: branch on it wave-uniformly, so that the reciprocal
         * is evaluated only when contact c really slides */
        const bool zero = fnorm < RKFD_DEV_TOL || fs < RKFD_DEV_TOL;
        double n1 = zero ? 0.0 : ff0, n2 = zero ? 0.0 : ff1;
        if( ( BALLOT( !zero && fnorm > fs ) >> c ) & 1ull ){
          const double sc = fs*RKFD_RCP( fnorm );
          n1 = ff0*sc; n2 = ff1*sc;
        }
        const double d1 = BCAST( n1 - f1, c ), d2 = BCAST( n2 - f2, c );
        if( lane == c ){ f1 = n1; f2 = n2; }
        rn = fma( A0[3*c+1], d1, fma( A0[3*c+2], d2, rn ) );
        r1 = fma( A1[3*c+1], d1, fma( A1[3*c+2], d2, r1 ) );
        r2 = fma( A2[3*c+1], d1, fma( A2[3*c+2], d2, r2 ) );
      }
    }
  }
}

/* ------------------------------------------------------------------------ */
/* MLCP rigid branch (reference src/rkfd_mlcp.c:287-297).  Preconditions: sweep 2 and sweep 3
 * have been run with the wrenches applied so far (rkFDUpdateAccBias), so AC holds the free
 * accelerations and U / MS / CHOL hold U, 1/D and the factor of a float joint's Ia.
 *
 * The reference builds the contact-space matrix A column by column (unit force at a contact,
 * rkFDChainUpdateCachedABIPair, read the relative accelerations, src/rkfd_mlcp.c:76-122) and,
 * once the forces are found, re-runs the cached-ABI sweeps with them applied.  Here both use
 * the factorisation the sweeps already hold, H^-1 = (1-HpsiK)' D^-1 (1-HpsiK): a probe walks
 * from its contact link up to the root once, leaving the innovation nu_k(j) = -S_j' dp it
 * causes at every joint j it passes (scaled by sqrt(1/D_j); for a float joint the six
 * components of L^-1 dp).  Then
 *     A(r,k)   = sum over the joints common to both paths of nu_r(j) nu_k(j)      (+ relaxation),
 *     delta qdd = the sweep-3 recursion driven by sum_k f_k nu_k                  (rkfd_phase_sweep3<true>),
 * i.e. no per-column response walks and no second backward sweep; A comes out exactly
 * symmetric.  Output: contact forces CF, committed contact state, and the inputs of the delta
 * sweep (MS slot 1, U slot of float joints). */
template<bool prof> RKFD_DEV void rkfd_phase_mlcp(const rkfdDevModel &m, const rkfdLds &L, const double *bv, unsigned long long *pc)
{
  unsigned long long q0 = prof ? RKFD_CLOCK() : 0ull, q1;
#define MST(k) do{ if( prof ){ q1 = RKFD_CLOCK(); pc[k] += q1 - q0; q0 = q1; } }while(0)
  const int lane = LANE();
  const int nc = L.cnt[CNT_NRG];
  const int M = 3*nc;
  const int ld = M+1;
  const int NLV = m.nlevel, NL = m.nlink, NR = m.npurow;
  const int NSD = m.nside;
  const int PUS = NR*M;                               /* stride between the two sides of PU */
  const unsigned char *TOP = L.PL + NL*NLV;           /* where a force on a link stops propagating (255: static) */
  const unsigned char *FSL = TOP + NL;                /* float slot of a link */
  const unsigned char *FLK = FSL + NL;                /* link of a float slot */
  const double dt = m.dt;

  /* b: free relative acceleration, then *dt + relative velocity + compensation
   * (_rkFDSolverBiasAcc / BiasVel / RelaxationCompensation, reference src/rkfd_mlcp.c:58-74,146-188) */
  if( lane < nc ){
    const int j = L.lrg[lane], cinf = L.CIp[j], ci = RKFD_CI_CI( cinf );
    const int la = RKFD_CI_A( cinf ), lb = RKFD_CI_B( cinf );
    const double x[3] = { L.CX[3*L.asl[j]], L.CX[3*L.asl[j]+1], L.CX[3*L.asl[j]+2] };
    double ta[3], tb[3], ra[3], d[3];
    /* spatial-acceleration part of the point accelerations: a_O + alpha x p (the velocity-product
     * part and the relative velocity come from rkfd_phase_bvel) */
    d_cross( &L.AC[6*la], x, ta ); d_cross( &L.AC[6*lb], x, tb );
#pragma unroll
    for( int k=0; k<3; k++ ){
      ra[k] = ( L.AC[6*la+3+k] + ta[k] ) - ( L.AC[6*lb+3+k] + tb[k] );
      d[k] = x[k]-L.RW[3*L.asl[j]+k];
    }
    const double mu = L.typ[j] == RKFD_SF ? m.ci_sf[ci] : m.ci_kf[ci];
    const double K = m.ci_k[ci];
#pragma unroll
    for( int i=0; i<3; i++ ){
      const double *ax = &L.AX[9*L.asl[j]+3*i];
      double b = ( d_dot( ax, ra ) + bv[3+i] )*dt + bv[i];
      b += ( i == 0 ? K : K*mu )*d_dot( d, ax );
      L.MB[3*lane+i] = b;
    }
    /* the moving side(s) of this contact, packed (RKFD_CS_*): link, its depth, the link where its
     * path ends (TOP), the first level of the path that carries a 1-DoF joint, float-top flag,
     * side.  One entry per contact when no rigid pair has two moving links, else one per side. */
    if( NSD == 1 ) L.tgt[lane] = 0;      /* (a contact between two immovable links has no moving side) */
#pragma unroll
    for( int sd=0; sd<2; sd++ ){
      const int a = sd == 0 ? la : lb;
      const int top = TOP[a];
      if( NSD == 1 && top == 255 ) continue;
      const int lit = L.LI[top == 255 ? 0 : top], jtt = RKFD_LI_JT( lit );
      const int d0 = RKFD_LI_DEPTH( lit ) + ( jtt == RKFD_JOINT_REVOL || jtt == RKFD_JOINT_PRISM ? 0 : 1 );
      const unsigned e = (unsigned)a | ( (unsigned)RKFD_LI_DEPTH( L.LI[a] ) << 8 ) | ( (unsigned)top << 14 ) | ( (unsigned)d0 << 22 )
                       | ( jtt == RKFD_JOINT_FLOAT ? 1u << 29 : 0u ) | ( (unsigned)sd << 30 ) | ( top != 255 ? 1u << 31 : 0u );
      L.tgt[NSD == 1 ? lane : 2*lane+sd] = (int)e;
    }
  }
  /* sqrt(1/D) of the 1-DoF joints (MS slot 2: the driving torque kept there is dead after sweep 2) */
  if( lane < NL ){
    const int jt = RKFD_LI_JT( L.LI[lane] );
    if( jt == RKFD_JOINT_REVOL || jt == RKFD_JOINT_PRISM ) L.MS[4*lane+2] = sqrt( L.MS[4*lane+0] );
  }
  SYNC();
  MST(14);
  /* probes: lane = column k = 3c+i; unit force along axis i at contact c, applied to the
   * owner link (+) and the other link (-).  Every level between the contact link and the top of
   * its path carries a 1-DoF joint; the operands of the next level are fetched while this one is
   * computed. */
  for( int cb=0; cb<M; cb+=RKFD_WAVE ){      /* 64 probe columns at a time */
    const int col = cb + lane;
    const bool on = col < M;
    const int c = on ? col/3 : 0, ia = on ? col%3 : 0;
    const int j = L.lrg[c];
    double W[6];
    {
      const double x[3] = { L.CX[3*L.asl[j]], L.CX[3*L.asl[j]+1], L.CX[3*L.asl[j]+2] };
      const double *ax = &L.AX[9*L.asl[j]+3*ia];
      d_cross( x, ax, W );
      W[3] = ax[0]; W[4] = ax[1]; W[5] = ax[2];
    }
    if( on ){
      for( int s2=0; s2<NSD; s2++ ){
        const unsigned e = (unsigned)L.tgt[c*NSD+s2];
        if( !RKFD_CS_VALID( e ) ) continue;
        const int a = RKFD_CS_LINK( e ), da = RKFD_CS_DEPTH( e ), d0 = RKFD_CS_D0( e );
        /* bias force delta: p = -f_ext */
        double dp[6];
        const double sg = RKFD_CS_SIDE( e ) == 0 ? -1.0 : 1.0;
#pragma unroll
        for( int k=0; k<6; k++ ) dp[k] = sg*W[k];
        double *pu = &L.PU[s2*PUS + col];
        const unsigned char *path = &L.PL[a*NLV];
        double Sx[6], Ux[6], sdx, dix;
#pragma unroll
        for( int k=0; k<6; k++ ){ Sx[k] = L.S[6*a+k]; Ux[k] = L.U[6*a+k]; }
        sdx = L.MS[4*a+2]; dix = L.MS[4*a+0];
        int inext = path[da > 0 ? da-1 : 0];
        for( int d=da; d>=d0; d-- ){
          const int in_ = inext;
          double Sn[6], Un[6];
#pragma unroll
          for( int k=0; k<6; k++ ){ Sn[k] = L.S[6*in_+k]; Un[k] = L.U[6*in_+k]; }
          const double sdn = L.MS[4*in_+2], din = L.MS[4*in_+0];
          inext = path[d > 1 ? d-2 : 0];
          double du0 = Sx[0]*dp[0], du1 = Sx[1]*dp[1];
          du0 = fma( Sx[2], dp[2], du0 ); du1 = fma( Sx[3], dp[3], du1 );
          du0 = fma( Sx[4], dp[4], du0 ); du1 = fma( Sx[5], dp[5], du1 );
          const double du = -( du0 + du1 );
          pu[d*M] = du*sdx;
          const double t = du*dix;
#pragma unroll
          for( int k=0; k<6; k++ ) dp[k] = fma( Ux[k], t, dp[k] );
#pragma unroll
          for( int k=0; k<6; k++ ){ Sx[k] = Sn[k]; Ux[k] = Un[k]; }
          sdx = sdn; dix = din;
        }
        if( RKFD_CS_FLOAT( e ) ){
          /* delta a = IA^-1 ( -dp ) = L^-T y,  y = L^-1 ( -dp ) */
          double rhs[6], y[6];
#pragma unroll
          for( int k=0; k<6; k++ ) rhs[k] = -dp[k];
          d_chol6_fwd( &L.CHOL[36*FSL[RKFD_CS_TOP( e )]], rhs, y );
#pragma unroll
          for( int k=0; k<6; k++ ) pu[( NLV+k )*M] = y[k];
        }
      }
    }
  }
  SYNC();
  MST(15);
  /* A, one 3x3 block per lane and pass: block ( cr, ck <= cr ) and its mirror image */
  for( int e0=0; e0<nc*nc; e0+=RKFD_WAVE ){
    const int e = e0 + lane;
    const int cr = e/nc, ck = e - cr*nc;
    if( e < nc*nc && ck <= cr ){
      double blk[9] = {0,0,0,0,0,0,0,0,0};
      for( int sr=0; sr<NSD; sr++ ) for( int sk=0; sk<NSD; sk++ ){
        const unsigned er = (unsigned)L.tgt[cr*NSD+sr], ek = (unsigned)L.tgt[ck*NSD+sk];
        if( !RKFD_CS_VALID( er ) || !RKFD_CS_VALID( ek ) || RKFD_CS_TOP( er ) != RKFD_CS_TOP( ek ) ) continue;   /* no joint in common */
        const double *pr = &L.PU[sr*PUS + 3*cr], *pk = &L.PU[sk*PUS + 3*ck];
        const int a = RKFD_CS_LINK( er ), b = RKFD_CS_LINK( ek );
        const int d0 = RKFD_CS_D0( er );
        int dc = RKFD_CS_DEPTH( er ) < RKFD_CS_DEPTH( ek ) ? RKFD_CS_DEPTH( er ) : RKFD_CS_DEPTH( ek );
        if( a != b ){
          /* last level the two paths share */
          int d = d0;
          while( d <= dc && L.PL[a*NLV+d] == L.PL[b*NLV+d] ) d++;
          dc = d-1;
        }
#pragma unroll 2
        for( int d=d0; d<=dc; d++ ){
          const double r0 = pr[d*M], r1 = pr[d*M+1], r2 = pr[d*M+2];
          const double k0 = pk[d*M], k1 = pk[d*M+1], k2 = pk[d*M+2];
          blk[0] = fma( r0, k0, blk[0] ); blk[1] = fma( r0, k1, blk[1] ); blk[2] = fma( r0, k2, blk[2] );
          blk[3] = fma( r1, k0, blk[3] ); blk[4] = fma( r1, k1, blk[4] ); blk[5] = fma( r1, k2, blk[5] );
          blk[6] = fma( r2, k0, blk[6] ); blk[7] = fma( r2, k1, blk[7] ); blk[8] = fma( r2, k2, blk[8] );
        }
        if( RKFD_CS_FLOAT( er ) ){
#pragma unroll
          for( int q=0; q<6; q++ ){
            const int d = NLV + q;
            const double r0 = pr[d*M], r1 = pr[d*M+1], r2 = pr[d*M+2];
            const double k0 = pk[d*M], k1 = pk[d*M+1], k2 = pk[d*M+2];
            blk[0] = fma( r0, k0, blk[0] ); blk[1] = fma( r0, k1, blk[1] ); blk[2] = fma( r0, k2, blk[2] );
            blk[3] = fma( r1, k0, blk[3] ); blk[4] = fma( r1, k1, blk[4] ); blk[5] = fma( r1, k2, blk[5] );
            blk[6] = fma( r2, k0, blk[6] ); blk[7] = fma( r2, k1, blk[7] ); blk[8] = fma( r2, k2, blk[8] );
          }
        }
      }
      if( cr == ck ){
        /* relaxation on the diagonal */
        const double rl = m.ci_l[RKFD_CI_CI( L.CIp[L.lrg[cr]] )];
        blk[0] += rl; blk[4] += rl; blk[8] += rl;
      }
#pragma unroll
      for( int i=0; i<3; i++ )
#pragma unroll
        for( int q=0; q<3; q++ ){
          L.MA[( 3*cr+i )*ld + 3*ck+q] = blk[3*i+q];
          if( cr != ck ) L.MA[( 3*ck+q )*ld + 3*cr+i] = blk[3*i+q];
        }
    }
  }
  SYNC();
  MST(6);
  /* projected Gauss-Seidel, fixed max_iter sweeps, no warm start (_rkFDSolverMLCP, reference
   * src/rkfd_mlcp.c:190-249), same update order.  lane = contact: each lane keeps the three
   * residuals res = b + A f, forces and inverse diagonals of ITS contact in registers, every lane
   * evaluates its own Gauss-Seidel candidate, and only the increment of the contact whose turn it
   * is gets broadcast (v_readlane) and applied to everybody's residuals. */
  {
    const bool on = lane < nc;
    const int r0 = on ? 3*lane : 0;
    double rn = 0, r1 = 0, r2 = 0, fn = 0, f1 = 0, f2 = 0, in_ = 0, i1 = 0, i2 = 0, mu = 0;
    if( on ){
      rn = L.MB[r0]; r1 = L.MB[r0+1]; r2 = L.MB[r0+2];
      const double dn = L.MA[r0*ld+r0], d1 = L.MA[(r0+1)*ld+r0+1], d2 = L.MA[(r0+2)*ld+r0+2];
      in_ = 1.0/dn;
      /* tangential rows with |a_kk| < zTOL are frozen at 0 (reference :220-221) */
      i1 = fabs( d1 ) < RKFD_DEV_TOL ? 0.0 : 1.0/d1;
      i2 = fabs( d2 ) < RKFD_DEV_TOL ? 0.0 : 1.0/d2;
      const int jr_ = L.lrg[lane], cir_ = RKFD_CI_CI( L.CIp[jr_] );
      mu = L.typ[jr_] == RKFD_SF ? m.ci_sf[cir_] : m.ci_kf[cir_];
    }
    const double *Arow = &L.MA[r0*ld];
    if( nc <= RKFD_PGS_NC ) rkfd_pgs_registers( Arow, ld, nc, m.max_iter, on, lane, mu, in_, i1, i2, rn, r1, r2, fn, f1, f2 );
    else for( int it=0; it<m.max_iter; it++ ){
      for( int c=0; c<nc; c++ ){
        /* normal force of contact c: f_n <- max( 0, -( b + a.f - a_nn f_n ) / a_nn ) */
        const double a0 = Arow[3*c], a1 = Arow[ld+3*c], a2 = Arow[2*ld+3*c];
        double ff = fn - rn*in_;
        if( ff < RKFD_DEV_TOL ) ff = 0.0;
        const double dl = BCAST( ff - fn, c );
        if( lane == c ) fn = ff;
        rn = fma( a0, dl, rn ); r1 = fma( a1, dl, r1 ); r2 = fma( a2, dl, r2 );
      }
      for( int c=0; c<nc; c++ ){
        /* tangential forces of contact c: Gauss-Seidel value for both, then scaled onto the
         * friction disc of radius mu f_n */
        const double a0 = Arow[3*c+1], a1 = Arow[ld+3*c+1], a2 = Arow[2*ld+3*c+1];
        const double b0 = Arow[3*c+2], b1 = Arow[ld+3*c+2], b2 = Arow[2*ld+3*c+2];
        const double ff0 = f1 - r1*i1, ff1 = f2 - r2*i2;
        const double fnorm = ff0*ff0 + ff1*ff1;
        double fs = mu*fn; fs = fs*fs;
        double n1, n2;
        if( fnorm < RKFD_DEV_TOL || fs < RKFD_DEV_TOL ){ n1 = 0; n2 = 0; }
        else if( fnorm > fs ){ const double sc = fs*RKFD_RCP( fnorm ); n1 = ff0*sc; n2 = ff1*sc; }
        else { n1 = ff0; n2 = ff1; }
        const double d1 = BCAST( n1 - f1, c ), d2 = BCAST( n2 - f2, c );
        if( lane == c ){ f1 = n1; f2 = n2; }
        rn = fma( a0, d1, fma( b0, d2, rn ) );
        r1 = fma( a1, d1, fma( b1, d2, r1 ) );
        r2 = fma( a2, d1, fma( b2, d2, r2 ) );
      }
    }
    if( on ){ L.MF[r0] = fn/dt; L.MF[r0+1] = f1/dt; L.MF[r0+2] = f2/dt; }
  }
  SYNC();
  MST(21);
  /* _rkFDSolverSetForce (reference src/rkfd_mlcp.c:252-284) incl. quirks Q1 / Q2 */
  if( lane < nc ){
    const int j = L.lrg[lane], ci = RKFD_CI_CI( L.CIp[j] );
    double fw[3] = {0,0,0};
#pragma unroll
    for( int i=0; i<3; i++ ){
      const double fi = L.MF[3*lane+i];
      fw[0] += fi*L.AX[9*L.asl[j]+3*i]; fw[1] += fi*L.AX[9*L.asl[j]+3*i+1]; fw[2] += fi*L.AX[9*L.asl[j]+3*i+2];
    }
    { const int sl_ = L.asl[j]; L.CF[3*sl_] = fw[0]; L.CF[3*sl_+1] = fw[1]; L.CF[3*sl_+2] = fw[2]; }
    const double fn = fw[0], fs = sqrt( fw[1]*fw[1] + fw[2]*fw[2] );
    const double mu = L.typ[j] == RKFD_SF ? m.ci_sf[ci] : m.ci_kf[ci];
    if( fs > mu*fn - RKFD_DEV_TOL ){
      L.typ[j] = RKFD_KF;
      { const int ri = RIDX( j ), sl_ = L.asl[j]; L.REF[3*ri] = L.PRO[3*sl_]; L.REF[3*ri+1] = L.PRO[3*sl_+1]; L.REF[3*ri+2] = L.PRO[3*sl_+2]; }
    } else {
      L.typ[j] = RKFD_SF;
    }
  }
  SYNC();
  MST(22);
  /* inputs of the delta sweep, lane = link: what the solved forces F = MF do to the joint's
   * innovation.  1-DoF joint: sum_k F_k nu_k / D  (= scaled sum times sqrt(1/D));  float joint:
   * sum_k F_k y_k.  Links no contact path passes get 0. */
  {
    const int ntask = NL + 6*m.nfloat;    /* one per link (used by those with a 1-DoF joint), then six per float joint */
    for( int t0=0; t0<ntask; t0+=RKFD_WAVE ){
      const int t = t0 + lane;
      const bool isl = t < NL, isf = !isl && t < ntask;
      const int fq = isf ? ( t-NL )%6 : 0;
      const int link = isl ? t : ( isf ? FLK[( t-NL )/6] : 0 );
      const int lii = L.LI[link], jt = RKFD_LI_JT( lii );
      const bool is1 = isl && ( jt == RKFD_JOINT_REVOL || jt == RKFD_JOINT_PRISM );
      const int dpt = isl ? RKFD_LI_DEPTH( lii ) : 0;
      const int row = isf ? NLV+fq : dpt;
      double sum = 0;
#pragma unroll 2
      for( int cs=0; cs<nc*NSD; cs++ ){
        const unsigned e = (unsigned)L.tgt[cs];
        const int c = NSD == 1 ? cs : cs >> 1;
        const double *pu = &L.PU[( NSD == 1 ? 0 : ( cs & 1 ) )*PUS + row*M + 3*c];
        const double v = L.MF[3*c]*pu[0] + L.MF[3*c+1]*pu[1] + L.MF[3*c+2]*pu[2];
        /* 1-DoF joint: it lies on the moving path of the contact side; float joint: the path ends there */
        const bool onp = RKFD_CS_VALID( e ) && ( isf ? RKFD_CS_TOP( e ) == link
                       : ( RKFD_CS_DEPTH( e ) >= dpt && RKFD_CS_D0( e ) <= dpt && L.PL[RKFD_CS_LINK( e )*NLV+dpt] == link ) );
        sum += onp ? v : 0.0;
      }
      if( is1 ) L.MS[4*link+1] = sum*L.MS[4*link+2];
      if( isf ) L.U[6*link+fq] = sum;
    }
  }
  SYNC();
  MST(23);
#undef MST
}

/* ------------------------------------------------------------------------ */
/* one dynamics evaluation: _rkFDUpdate / _rkFDUpdateRef (reference src/rkfd_sim.c:533-549).
 * Input L.q, L.qd; output L.acc (and contact / pivot state).  Returns nonzero when the model
 * needs a rigid solver that is not available on the device (wave-uniform). */
template<bool prof> RKFD_DEV int rkfd_evaluate(const rkfdDevModel &m, const rkfdLds &L, rkfdLaneLink &ll, bool doUpRef, unsigned long long *pc)
{
  const int lane = LANE();
  int err = 0;
  unsigned long long t0 = 0, t1;
#define STAMP(k) do{ if( prof ){ t1 = RKFD_CLOCK(); pc[k] += t1 - t0; t0 = t1; } }while(0)
  if( prof ) t0 = RKFD_CLOCK();
  if( lane < m.ndof ) L.acc[lane] = 0.0;
  rkfd_phase_kinematics<prof>( m, L, ll, pc );
  STAMP(0);
  /* commit joint friction pivots (the reference does so inside rkFDJointFrictionRevolDC) */
  if( doUpRef && lane < m.nlink ){
    const int jt = RKFD_LI_JT( L.LI[lane] );
    if( ( jt == RKFD_JOINT_REVOL || jt == RKFD_JOINT_PRISM ) && RKFD_LI_MT( L.LI[lane] ) == RKFD_MOTOR_DC )
      ll.pivt = (int)L.MS[4*lane+1];
  }
  /* MS slot 0 carries (driving torque + friction) until sweep 2 overwrites it: keep a copy */
  double drv = 0;
  if( lane < m.nlink ) drv = L.MS[4*lane+0];
  SYNC();
  if( m.ncand > 0 ){
    rkfd_phase_collision( m, L );
    if( L.cnt[CNT_NEL] > 0 ) rkfd_phase_penalty( m, L, doUpRef );
  } else if( lane == 0 ){
    L.cnt[CNT_NRG] = 0; L.cnt[CNT_NEL] = 0;
  }
  SYNC();
  double bv[6];
  rkfd_phase_bvel( m, L, bv );
  STAMP(1);
  /* rkChainUpdateABI (with rigid contacts this is rkFDUpdateAccBias) */
  rkfd_phase_sweep2<prof>( m, L, pc );
  STAMP(2);
  rkfd_phase_sweep3<false>( m, L );
  STAMP(3);
  if( L.cnt[CNT_NRG] > 0 ){
    if( m.solver == RKFD_SOLVER_MLCP ){
      /* contact forces, then their effect on the accelerations (rkChainUpdateCachedABI in the reference) */
      rkfd_phase_mlcp<prof>( m, L, bv, pc );
      STAMP(4);
      rkfd_phase_sweep3<true>( m, L );
      STAMP(3);
    } else {
      err = 1;
    }
  }
  /* rkFDUpdateJointPrevDrivingTrq (reference src/rkfd_util.c:289-311), committing evaluation only */
  if( doUpRef && lane < m.nlink ){
    const int jt = RKFD_LI_JT( L.LI[lane] );
    if( jt == RKFD_JOINT_REVOL || jt == RKFD_JOINT_PRISM )
      ll.pivp = drv - L.MS[4*lane+3]*L.acc[RKFD_LI_OFF( L.LI[lane] )];
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
template<bool prof> RKFD_DEV void rkfd_instance(const rkfdDevModel &m, const rkfdDevState &st, int b, void *ldsbase,
                            int mode, int nsteps, int *errflag)
{
  const int lane = LANE();
  const int ND = m.ndof, NL = m.nlink, NC = m.ncand;
  rkfdLds L;
  rkfd_lds_carve( &L, ldsbase, NL, ND, NC, 3*m.maxrg, m.nlevel, m.npool, m.nfloat, m.maxact, m.nside, m.pu_alias, m.npurow );
  if( lane == 0 ) L.cnt[CNT_OVF] = 0;

  /* load persistent state */
  double q = 0, qd = 0;
  int dofkind = 0;      /* 1: first angular coordinate of a float joint, 2: the other two */
  if( lane < ND ){ q = st.dis[(size_t)b*ND+lane]; qd = st.vel[(size_t)b*ND+lane]; dofkind = m.dofkind[lane]; }
  rkfdLaneLink ll; ll.min = 0; ll.pivp = 0; ll.pivt = 0;
  if( lane < NL ){
    L.LI[lane]   = m.linfo[lane];
    L.CHI[lane]  = m.child_idx[lane];
    L.PSL[lane]  = m.pslot[lane];
    const int lm = m.orig[lane];
    ll.min  = st.motor_in[(size_t)b*m.nlink_model+lm];
    ll.pivt = st.piv_type[(size_t)b*m.nlink_model+lm];
    ll.pivp = st.piv_prev[(size_t)b*m.nlink_model+lm];
  }
  if( m.maxrg > 0 ){
    for( int k=lane; k<NL*( m.nlevel+3 ); k+=RKFD_WAVE ) L.PL[k] = (unsigned char)m.pathlink[k];
  }
  for( int c0=0, base=0; c0<NC; c0+=RKFD_WAVE ){
    const int j = c0 + lane;
    const bool onj = j < NC;
    int a = 0;
    if( onj ){
      L.CIp[j] = m.cinfo[j];
      L.CFO[j] = m.cand_foff[j];
      a = st.cv_active[(size_t)b*NC+j];
      L.typ[j] = a ? st.cv_type[(size_t)b*NC+j] : 0;
    }
    /* slots of the contacts alive at launch (candidate order; re-assigned by every collision pass) */
    const unsigned long long ma = BALLOT( a != 0 );
    const int sl = base + __builtin_popcountll( ma & ( lane == 0 ? 0ull : ( ~0ull >> ( 64-lane ) ) ) );
    base += __builtin_popcountll( ma );
    if( a && sl >= m.maxact ) a = 0;
    if( onj ){
      L.act[j] = a;
      L.asl[j] = a ? sl : 0;
      if( a ){
#pragma unroll
        for( int k=0; k<3; k++ ) L.REF[3*sl+k] = st.cv_ref[((size_t)b*NC+j)*3+k];
      }
    }
  }
  SYNC();
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
      err |= rkfd_evaluate<prof>( m, L, ll, doUp, pc );
      const double a = on ? L.acc[lane] : 0.0;
      if( stage == 0 ){ Fv = xv; Fa = a; Tv = xv; Ta = a; Pv = c21*xv; Pa = c21*a; }
      else if( stage == 1 ){ Fv = fma( w2, xv, Fv ); Fa = fma( w2, a, Fa ); Tv = fma( c22, xv, Pv ); Ta = fma( c22, a, Pa ); Pv = c31*xv; Pa = c31*a; }
      else if( stage == 2 ){ Fv = fma( w3, xv, Fv ); Fa = fma( w3, a, Fa ); Tv = fma( c32, xv, Pv ); Ta = fma( c32, a, Pa ); }
      else if( stage == 3 ){ Fv += xv; Fa += a; Tv = Fv; Ta = Fa; }
      SYNC();
      stage++; if( stage == nst ) stage = 0;
    }
  }
  if( prof && lane == 0 && st.prof ){
    pc[prof ? 7 : 0] = RKFD_CLOCK() - tstart;
#pragma unroll
    for( int k=0; k<( prof ? RKFD_NPROF : 1 ); k++ ) st.prof[(size_t)b*RKFD_NPROF+k] = pc[k];
  }
  /* store */
  if( lane < ND ){
    st.dis[(size_t)b*ND+lane] = q; st.vel[(size_t)b*ND+lane] = qd;
    st.acc[(size_t)b*ND+lane] = L.acc[lane];
  }
  if( lane < NL ){
    const int lm = m.orig[lane];
    st.piv_type[(size_t)b*m.nlink_model+lm] = ll.pivt;
    st.piv_prev[(size_t)b*m.nlink_model+lm] = ll.pivp;
  }
  /* contact state: the flag of every candidate, the rest only for those in contact (a candidate out of
   * contact has no state: type and anchor are re-initialised at its next first contact, and the
   * boundary reports zeros for it) */
  for( int j=lane; j<NC; j+=RKFD_WAVE ){
    const int a = L.act[j];
    st.cv_active[(size_t)b*NC+j] = a;
    if( a ){
      st.cv_type[(size_t)b*NC+j] = L.typ[j];
#pragma unroll
      for( int k=0; k<3; k++ ){
        st.cv_ref[((size_t)b*NC+j)*3+k] = L.REF[3*RIDX( j )+k];
        st.cv_f[((size_t)b*NC+j)*3+k] = L.CF[3*L.asl[j]+k];
      }
    }
  }
  if( st.dbg ){
    /* debug dump: spatial accelerations (6/link) */
    if( lane < NL ){
      double *o = st.dbg + (size_t)b*st.dbg_stride;
      for( int k=0; k<6; k++ ) o[6*lane+k] = L.AC[6*lane+k];
    }
  }
  if( lane == 0 && errflag ){
    if( err ) *errflag = 1;              /* rigid contact with a solver that has no device path */
    if( L.cnt[CNT_OVF] ) *errflag = 2;   /* more rigid contacts than the configured capacity   */
  }
}

#endif /* RKFD_DEVICE_H */
