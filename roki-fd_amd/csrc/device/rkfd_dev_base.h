/* rkfd_dev_base.h - part of the device code of the batched rkFDUpdate step (see rkfd_device.h):
 * platform macros (HIP / lane emulator), cross-lane helpers, small vector algebra, the LDS carve-up.
 * Included by rkfd_device.h only, in this order; compiles for gfx950 and under the lane emulator. */
#ifndef RKFD_DEV_BASE_H
#define RKFD_DEV_BASE_H

#ifdef RKFD_EMU
#  define RKFD_DEV static inline
   int    rkfd_emu_lane(void);
   void   rkfd_emu_sync(void);
   void   rkfd_emu_sync_wave(void);
   double rkfd_emu_g8sum(double x);
   double rkfd_emu_bcast(double x, int src);
   unsigned long long rkfd_emu_ballot(int pred);
#  define LANE()        rkfd_emu_lane()
   int    rkfd_emu_half(void);
#  define HALF()        rkfd_emu_half()
#  define SYNC()        rkfd_emu_sync()
#  define SYNCW()       rkfd_emu_sync_wave()      /* all live instances of the wavefront (two instances per wavefront: the shared tables) */
   double rkfd_emu_g8bcast(double x, int k);
#  define G8SUM(x)      rkfd_emu_g8sum(x)
#  define G8SUM2(x,y)   do{ (x) = rkfd_emu_g8sum(x); (y) = rkfd_emu_g8sum(y); }while(0)
#  define G8BCAST(x,k)  rkfd_emu_g8bcast(x,k)
#  define RKFD_RCP(x)   ( 1.0/(x) )
#  define LDS_FENCE()   rkfd_emu_sync()
#  define RKFD_SCHED_BARRIER() do{}while(0)
#  define BCAST(x,l)    rkfd_emu_bcast(x,l)
#  define BCASTI(x,l)   ( (int)rkfd_emu_bcast( (double)(x), l ) )
#  define BALLOT(p)     rkfd_emu_ballot(p)
#  define ANY(p)        ( rkfd_emu_ballot(p) != 0ull )
   double rkfd_emu_wsum(double x);
   double rkfd_emu_wmin(double x);
#  define WSUM(x)       rkfd_emu_wsum(x)
#  define WMIN(x)       rkfd_emu_wmin(x)
#  define ROWBC_FMAC(C,acc,x,a) ( (acc) = fma( rkfd_emu_bcast( (x), ( rkfd_emu_lane() & ~15 ) | (C) ), (a), (acc) ) )
#else
#  define RKFD_DEV __device__ __forceinline__
/* the lane index is read through an opaque asm in every phase: otherwise the compiler hoists dozens of
 * lane-derived addresses out of the step loop (loop-invariant code motion) and keeps them in registers
 * across all phases - ~100 VGPRs of the whole-kernel pressure that no single phase needs */
#if RKFD_W == 1
static __device__ __forceinline__ int rkfd_lane(void){ int l = (int)threadIdx.x; asm volatile( "" : "+v"(l) ); return l; }
#  define HALF()        0
#else
/* two instances per wavefront: the lane within the instance's half, and which half */
static __device__ __forceinline__ int rkfd_lane(void){ int l = (int)threadIdx.x & ( RKFD_WL-1 ); asm volatile( "" : "+v"(l) ); return l; }
#  define HALF()        ( (int)threadIdx.x >> 5 )
#endif
#  define LANE()        rkfd_lane()
/* One workgroup is one wavefront: lanes exchange data through LDS in program order, so a
 * "barrier" only has to (a) stop the compiler from moving LDS accesses across it and (b) wait
 * for the wave's own outstanding LDS operations.  __syncthreads() would also drain vmcnt (the
 * schedule-record prefetches), which is exactly the latency the prefetch is meant to hide. */
#  define SYNC()        asm volatile( "s_waitcnt lgkmcnt(0)" ::: "memory" )
#  define SYNCW()       SYNC()      /* (the instances of a wavefront run in lockstep: what one half wrote in program order the other reads) */
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
#if RKFD_W == 1
/* broadcast lane src (wave-uniform) to every lane */
RKFD_DEV double rkfd_bcast(double x, int src)
{
  int lo = __builtin_amdgcn_readlane( __double2loint( x ), src );
  int hi = __builtin_amdgcn_readlane( __double2hiint( x ), src );
  return __hiloint2double( hi, lo );
}
#else
/* two instances per wavefront: lane src of the OWN half to every lane of the half (src the same within a half; the crossbar of
 * ds_bpermute instead of a scalar register, which holds one value per wavefront) */
RKFD_DEV int rkfd_bcasti(int x, int src){ return __builtin_amdgcn_ds_bpermute( ( ( (int)threadIdx.x & 32 ) + src ) << 2, x ); }
RKFD_DEV double rkfd_bcast(double x, int src)
{
  int lo = rkfd_bcasti( __double2loint( x ), src );
  int hi = rkfd_bcasti( __double2hiint( x ), src );
  return __hiloint2double( hi, lo );
}
#endif
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
/* acc += x[lane C of this lane's row of 16] * a in ONE instruction: gfx90a+ gives the fp64 FMA a DPP operand with
 * row_newbcast, so a Gauss-Seidel increment reaches the residuals of every lane of the row without a trip through
 * the scalar registers (v_readlane pair + use: ~37 cycles measured).  C is a literal 0..15.  The s_nop covers the
 * VALU-write -> DPP-read hazard, which the compiler does not see inside inline asm. */
template<int C> RKFD_DEV void rkfd_rowbc_fmac(double &acc, double x, double a)
{
  asm volatile( "s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(a), "n"(C) );
}
#  define ROWBC_FMAC(C,acc,x,a) rkfd_rowbc_fmac<C>( acc, x, a )
#  define G8SUM(x)      rkfd_g8sum(x)
#  define G8SUM2(x,y)   rkfd_g8sum2(x,y)
#  define G8BCAST(x,k)  rkfd_g8bcast<k>(x)
#  define RKFD_RCP(x)   rkfd_rcp(x)
/* compiler-only fence: LDS operations of one wavefront execute in program order */
#  define LDS_FENCE()   asm volatile( "" ::: "memory" )
/* the instruction scheduler does not move anything across this point */
#  define RKFD_SCHED_BARRIER() __builtin_amdgcn_sched_barrier( 0 )
#  define BCAST(x,l)    rkfd_bcast(x,l)
/* does any lane of the WAVEFRONT vote yes: a wave-uniform branch condition whatever the number of instances in the wavefront */
#  define ANY(p)        ( __ballot(p) != 0ull )
#if RKFD_W == 1
#  define BCASTI(x,l)   __builtin_amdgcn_readlane( (int)(x), l )      /* an int of lane l (wave-uniform l) */
#  define BALLOT(p)     __ballot(p)
#else
#  define BCASTI(x,l)   rkfd_bcasti( (int)(x), l )
/* the votes of the instance's own half, in bits 0 .. 31 (a per-lane value: what is branched on it diverges between the halves,
 * which the compiler handles with the execution mask) */
#  define BALLOT(p)     ( ( __ballot(p) >> ( (int)threadIdx.x & 32 ) ) & 0xffffffffull )
#endif
/* sum / minimum over the whole wave in registers, the same value in every lane: the 8-lane DPP butterfly, then the eight
 * group results through v_readlane (no LDS, no barrier: ~100 cycles where a tree through LDS takes ~2000) */
RKFD_DEV double rkfd_wave_sum(double x)
{
  x = rkfd_g8sum( x );
  double r = rkfd_bcast( x, 0 );
#pragma unroll
  for( int g=1; g<8; g++ ) r += rkfd_bcast( x, 8*g );
  return r;
}
RKFD_DEV double rkfd_wave_min(double x)
{
  x = fmin( x, rkfd_dpp_xor1( x ) );
  x = fmin( x, rkfd_dpp_xor2( x ) );
  x = fmin( x, rkfd_dpp_hmirror( x ) );
  double r = rkfd_bcast( x, 0 );
#pragma unroll
  for( int g=1; g<8; g++ ) r = fmin( r, rkfd_bcast( x, 8*g ) );
  return r;
}
#  define WSUM(x)       rkfd_wave_sum(x)
#  define WMIN(x)       rkfd_wave_min(x)
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
#define RKFD_NPROF 32
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
  double *q, *qd;                 /* [ndof] each; live from the integrator's update to the first look of the next evaluation:
                                     ALIAS the head of PB|AC|C */
  double *acc;                    /* [ndof] joint accelerations: ALIASES the head of IST (dead after sweep 2) and so MA - the
                                     free accelerations wait in a lane register while the contact problem is solved */
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
  double *MS;                     /* [NL*3]  Dinv, u, tau (slots re-used, see the phases)   */
  double *IST;                    /* [NL*14] inertia staging: A = Iw + m(|r|^2 1 - r r') (xx,xy,xz,yy,yz,zz),
                                     +m r (3), -m r (3), m, 0: every entry of the 6x6 is one of these */
  double *POOL;                   /* [npool*36] Ia of links whose parent gathers through LDS */
  double *CHOL;                   /* [nfloat*21] articulated inertia / Cholesky factor of float joints: lower triangle, packed by rows */
  double *XF;                     /* [nfloat*12] float joints: world orientation of the joint-origin frame (9), link position (3) */
  double *CX, *AX, *RW, *PRO;     /* per ACTIVE contact slot (capacity maxact): 3, 6 (normal, first tangent; d_load_axes), 3, 3 */
  double *REF;                    /* stick anchors (state): per active slot              */
  double *RTMP;                   /* [maxact*3] copy of REF while the slots are re-assigned; only when ncand > 64 */
  double *CF;                     /* contact forces (output): per active slot              */
  double *SV, *SD;                /* slide mode only: relative slide velocity of the two cells (world), anchor drift of one
                                     committing evaluation (anchor frame); per active slot, 3 each */
  double *QL, *QW, *QV, *CR;      /* Vert QP (only when the world can have rigid contacts under the Vert plugin):
                                     [M(M+1)/2] Q / its Cholesky factor (packed lower triangle), [M*M] W = L^-1 C', [5M (+64)] vectors (+ reduction scratch
                                     unless it overlays the link accelerations), [3M] reduced rows */
  unsigned char *CRC;             /* [M] contact of a reduced constraint row */
  unsigned char *FS;              /* [maxact] the slot's CF holds a force (it shares storage with RW) */
  double *QG, *QY;                /* the wide form (vert_rigid == 3): pyramid rows [3 P maxrg], multipliers [P maxrg] */
  unsigned char *QA;              /* ... active flags [P maxrg], active faces per contact [maxrg] */
  double *MA, *MB, *MF, *PU;      /* contact problem: [ma_size] the matrix (ALIASES IST|POOL; full rows or a packed lower triangle, rkfd_ma_idx),
                                     [M] bias vector, [M] forces (ALIAS the bias vector in the PGS kernels), [nside*npurow*M] (ALIASES C|PA when it fits) */
  int *tgt, *cnt;
  unsigned short *lrg, *lel;      /* [maxact] candidates in rigid / elastic contact, in candidate order */
  unsigned char *act, *typ;       /* [NC] in contact, stick / slip type                   */
  unsigned char *asl;             /* [NC] active-contact slot of a candidate              */
  int *LI;                        /* [NL] packed link info (RKFD_LI_*)                    */
  int *CIp;                       /* [NC] packed candidate info                           */
  unsigned short *CFO;            /* [NC] first plane of the candidate's partner shape    */
  unsigned short *CHP;            /* [NL] children lists (CSR values; offsets in the schedule): child | ( its pool slot + 1 ) << 8 */
  unsigned char *PL;              /* [NL*nlevel] ancestor at depth d (MLCP only), one byte each */
  unsigned char *BRK;             /* [NL] worlds with breakable float joints only: 0 no such joint, 1 unbroken, 2 broken (rkfd_dev_brf.h) */
  /* Volume plugin (kernel variant sv == 2 only; rkfd_dev_volume.h): per colliding pair VD [np*48] and its contact-plane
   * conditions VPL [np*ncp*8]; face polygons VPOLY [nf*pv*3] and reduction scratch VRED [16*nf+16] of the collision phase, sharing
   * their storage with the solve's:
   * QP: VQL [n(n+1)/2] Q / its factor, VQW [n*mc], VS, VEV [mc*mc], VQV [5n + mc]; simplex workspace VLP (over the QP arrays); VI [np*2]
   * conditions of a pair, its model pair.  n = 6 np, mc = np ( 1 + ncp ). */
  double *VD, *VPL, *VPOLY, *VRED, *VQL, *VQW, *VS, *VEV, *VQV, *VLP;
  int *VI;
  /* grouped Gauss-Seidel (worlds with more than 16 rigid contact vertices): the layout of the last evaluation - per lane the two
   * moving trees of its contact (2 bytes), the position table (64 bytes), the row fills, the contact count (RKFD_GC_INTS ints) */
  int *GC;
} rkfdLds;
RKFD_DEV void rkfd_lds_carve(rkfdLds *L, void *base, int NL, int ND, int NC, int M, int nlevel, int npool, int nfloat, int maxact, int nside, int pu_alias, int npurow, int vert_rigid, int has_slide, int ma_size,
                             int vol_np, int vol_ncp, int vol_pv, int vol_nf, int pyramid, int has_pl, void *shared = 0)
/* must match the byte count computed in rkfd_devmodel.cpp.  shared != 0: the world's static tables live there (once per wavefront,
 * rkfdDevModel.lds_shared) instead of in the instance's own block */
{
  double *d = (double *)base;
  L->S = d; d += NL*6;
  L->V = d; L->U = d; L->tmp = d; d += NL*6;
  /* the joint coordinates and rates (2 ND <= 12 NL doubles) are read when an evaluation starts and written by the
   * integrator after it ended: they borrow the bias-force / acceleration block and the velocity-product block,
   * which the kinematics phase fills only after its last look at them */
  L->q = d; L->qd = d + ND;
  L->PB = d; L->AC = d; d += NL*6;
  L->C = d; d += NL*6;
  L->PA = d; L->XA = d; d += NL*6;
  L->MS = d; d += NL*3;
  {
    const int pool = 36*npool > 6*NL ? 36*npool : 6*NL;
    int stage = 14*NL + pool;
    L->IST = d; L->POOL = d + 14*NL; L->XB = d + 14*NL; L->MA = d;
    /* the joint accelerations leave sweep 3 through the head of the (by then dead) inertia staging; the contact
     * matrix overwrites them, so rkfd_evaluate carries them in a lane register across the contact phase */
    L->acc = d;
    /* the contact matrix: rows padded to an odd stride while there is room, i.e. unless every slot is taken (the Vert
     * QP keeps the padded layout throughout) */
    if( ma_size > stage ) stage = ma_size;
    d += stage;
  }
  L->CHOL = d; d += 21*nfloat; L->XF = d; d += 12*nfloat;
  L->CX = d; d += maxact*3; L->AX = d; d += maxact*6; L->RW = d; d += maxact*3; L->PRO = d; d += maxact*3;
  L->REF = d; d += maxact*3; L->RTMP = d; if( NC > RKFD_WAVE/2 ) d += maxact*3;      /* (more than one chunk of candidates at two instances per wavefront) */
  /* the world force of a contact is written after the last look at its anchor in world coordinates (RW: the penalty force and the
   * bias of the rigid system read it): they share storage, and FS says whether a slot holds a force yet (a contact nobody solved -
   * beyond the capacity, or a rigid vertex under the Volume plugin - reports zero) */
  L->CF = L->RW;
  L->SV = d; L->SD = d; if( has_slide ){ L->SV = d; d += maxact*3; L->SD = d; d += maxact*3; }
  /* PGS: a lane reads its three entries of b before it writes its three forces, so they share storage */
  L->MB = d; d += M; L->MF = L->MB; if( vert_rigid ){ L->MF = d; d += M; }      /* (Volume plugin: the forces are written when the bias is long dead) */
  /* probe scratch: lives while the contact problem is set up and solved, when C and PA are dead */
  if( pu_alias ) L->PU = L->C; else { L->PU = d; d += nside*npurow*M; }
  L->QL = d; L->QW = d; L->QV = d; L->CR = d;
  if( vert_rigid ){ L->QL = d; d += M*( M+1 )/2; L->QW = d; if( vert_rigid != 2 ) d += M*M; L->QV = d; d += 5*M; L->CR = d; d += 3*M; }
  L->QG = d; L->QY = d; if( vert_rigid == 3 ){ L->QG = d; d += pyramid*M; L->QY = d; d += pyramid*( M/3 ); }
  L->VD = d; L->VPL = d; L->VPOLY = d; L->VRED = d; L->VQL = d; L->VQW = d; L->VS = d; L->VEV = d; L->VQV = d; L->VLP = d;
  if( vol_np ){
    const int n = 6*vol_np, mc = vol_np*( 1+vol_ncp );
    L->VD = d; d += vol_np*48; L->VPL = d; d += vol_np*vol_ncp*8;
    /* the scratch of the collision phase and that of the solve share their storage */
    L->VPOLY = d; L->VRED = d + vol_nf*vol_pv*3;
    double *e = d;
    L->VQL = e; e += n*( n+1 )/2; L->VQW = e; e += n*mc; L->VS = e; e += mc*mc; L->VEV = e; e += mc*mc;
    L->VQV = e; e += 5*n + mc; L->VLP = d;      /* (the simplex's workspace overlays the QP's arrays, dead by then) */
    d += RKFD_VOL_LDS_COL( vol_nf, vol_pv ) > RKFD_VOL_LDS_SOL( vol_np, vol_ncp, pyramid ) ? RKFD_VOL_LDS_COL( vol_nf, vol_pv ) : RKFD_VOL_LDS_SOL( vol_np, vol_ncp, pyramid );
  }
  int *ip = (int *)d;
  int *sip = (int *)shared;
  if( shared ){ L->CIp = sip; sip += NC; } else { L->CIp = ip; ip += NC; }
  L->tgt = ip; ip += nside*maxact; L->cnt = ip; ip += vol_np ? 12 : ( NC > 0 ? 8 : 4 );
  L->VI = ip; if( vol_np ) ip += 2*vol_np;
  L->GC = ip; if( RKFD_GC_NEEDED( M ) ) ip += RKFD_GC_INTS;
  if( shared ){ L->LI = sip; sip += NL; } else { L->LI = ip; ip += NL; }
  unsigned short *sp = (unsigned short *)ip;
  unsigned short *ssp = (unsigned short *)sip;
  if( shared ){ L->CHP = ssp; ssp += NL; L->CFO = ssp; ssp += NC; } else { L->CHP = sp; sp += NL; L->CFO = sp; sp += NC; }
  L->lrg = sp; sp += maxact; L->lel = sp; sp += maxact;
  unsigned char *bp = (unsigned char *)sp;
  L->act = bp; bp += NC; L->typ = bp; bp += NC; L->asl = bp; bp += NC;
  L->FS = bp; bp += maxact;
  L->CRC = bp; if( vert_rigid ) bp += M;
  L->QA = bp; if( vert_rigid == 3 ) bp += ( pyramid+1 )*( M/3 );
  if( shared ) L->PL = (unsigned char *)ssp; else { L->PL = bp; if( has_pl ) bp += NL*( nlevel+3 ); }
  L->BRK = bp;
}

/* the contact frame of a slot: normal and first tangent are stored, the second tangent is their cross product
 * (exactly how d_ortho_space made it) */
RKFD_DEV void d_load_axes(const rkfdLds &L, int slot, double *ax)
{
#pragma unroll
  for( int k=0; k<6; k++ ) ax[k] = L.AX[6*slot+k];
  d_cross( ax, ax+3, ax+6 );
}

/* per-lane state that only lane = link ever touches: kept in registers for the whole launch */
typedef struct { double min, pivp; int pivt;
  int dtask;   /* lane = coordinate: its link | component << 8 when the joint is a 1-DoF or a float joint, else -1 (the delta-sweep inputs) */
} rkfdLaneLink;

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

/* The probe scratch PU: entry ( side s, column col of the contact problem, tree level d ) - the scaled innovation the unit probe of
 * that column leaves at the joint of its path on level d; the six components of a float joint sit at levels nlevel .. nlevel + 5.
 * A column's levels are contiguous (the delta-sweep inputs read lane = level: neighbouring addresses), no 1-DoF joint sits above
 * level pu_d0, and every stride is a dimension of the WORLD (a literal in the kernels compiled for one world), not of the
 * evaluation's contact count. */
#define RKFD_PU_AT(m, s, col, d) ( ( (s)*3*(m).maxrg + (col) )*(m).npurow + (d) - (m).pu_d0 )

/* L->BRK: breakable float joints (rkfd_dev_brf.h) */
#define RKFD_BRF_NONE     0
#define RKFD_BRF_ATTACHED 1
#define RKFD_BRF_BROKEN   2

/* counters in L->cnt */
#define CNT_NRG 0
#define CNT_NEL 1
#define CNT_OVF 2
#define CNT_QPF 3
#define CNT_NTGT 4      /* (worlds with contact candidates only: four counters otherwise) */
#define CNT_SRG 5       /* running sums over the committing evaluations of this launch's steps: rigid contacts, */
#define CNT_SEL 6       /* elastic contacts, */
#define CNT_SN  7       /* steps (rkfdBatchContactStats) */
#define CNT_NVP 8       /* Volume plugin: rigid pairs in collision in this evaluation (twelve counters in that kernel variant) */
#define CNT_GRD 9       /* ... a guarded pair (a shape that cannot be clipped) was found in collision */

#endif /* RKFD_DEV_BASE_H */
