/* rkfd_dev_mlcp.h - part of the device code of the batched rkFDUpdate step (see rkfd_device.h):
 * phase: MLCP rigid contacts in innovations form + projected Gauss-Seidel.
 * Included by rkfd_device.h only, in this order; compiles for gfx950 and under the lane emulator. */
#ifndef RKFD_DEV_MLCP_H
#define RKFD_DEV_MLCP_H

/* projected Gauss-Seidel for up to RKFD_PGS_NC contacts with the lane's three matrix rows held in
 * registers (3 x 3*RKFD_PGS_NC doubles): the contact loop is unrolled, so the rows are indexed
 * statically, the broadcasts read fixed lanes and the dependent path of an update is ALU only.
 * Same arithmetic and update order as the general loop in rkfd_phase_mlcp. */
#define RKFD_PGS_NC 4
/* where entry (R, K) of the contact matrix lives: full rows (stride ld), or - kernel variants pk, chosen by the host
 * where it buys residency - the lower triangle packed by rows (A is exactly symmetric), which halves the largest
 * object in LDS (72 rows: 41 KB -> 21 KB, a third instance per CU for config 5).  The packed form costs index
 * arithmetic in the PGS (+12 % of that phase), so small matrices that fit anyway stay full. */
template<bool pk> RKFD_DEV int rkfd_ma_idx(int Rr, int K, int ld)
{
  if( !pk ) return Rr*ld + K;
  return K <= Rr ? ( Rr*( Rr+1 ) >> 1 ) + K : ( K*( K+1 ) >> 1 ) + Rr;
}
/* entry ( Rr, K ) as the Gauss-Seidel loops read it: from the TRANSPOSED position.  A is exactly symmetric (the matrix build
 * writes every off-diagonal block and its mirror image from the same registers, and a diagonal block's ( i, q ) and ( q, i ) are
 * the same products summed in the same order), so the value is bit for bit the same - but lane = contact reads rows
 * 3 lane, 3 lane + 1, 3 lane + 2: as ROWS of the full layout the lanes are one row stride x 3 apart (24 contacts x 3 = 72
 * doubles = 144 dwords when every slot is taken, the headline case: 16 banks apart, lanes 0 / 4 and 1 / 5 ... collide, counters:
 * SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 27 % on config 4); as positions r0 .. r0 + 2 of the column's row the
 * lanes are 3 doubles apart - no two in one bank - and a lane's three entries are neighbours in memory. */
template<bool pk> RKFD_DEV int rkfd_ma_idx_t(int Rr, int K, int ld){ return rkfd_ma_idx<pk>( K, Rr, ld ); }
template<bool pk> RKFD_DEV void rkfd_pgs_registers(const double *MA, int r0, int ld, int nc, int max_iter, bool on, int lane, double mu,
                                 double in_, double i1, double i2, double &rn, double &r1, double &r2, double &fn, double &f1, double &f2)
{
  double A0[3*RKFD_PGS_NC], A1[3*RKFD_PGS_NC], A2[3*RKFD_PGS_NC];
#pragma unroll
  for( int k=0; k<3*RKFD_PGS_NC; k++ ){
    const bool in = on && k < 3*nc;
    A0[k] = in ? MA[rkfd_ma_idx<pk>( r0, k, ld )] : 0.0; A1[k] = in ? MA[rkfd_ma_idx<pk>( r0+1, k, ld )] : 0.0; A2[k] = in ? MA[rkfd_ma_idx<pk>( r0+2, k, ld )] : 0.0;
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
        /* (a wave-uniform branch - does the contact whose turn it is slide, in any instance of the wavefront - and lane selects) */
        const bool sl = !zero && fnorm > fs;
        if( ANY( sl && lane == c ) ){
          const double sc = fs*RKFD_RCP( fnorm );
          n1 = sl ? ff0*sc : n1; n2 = sl ? ff1*sc : n2;
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

/* projected Gauss-Seidel for up to 16 contacts (lane = contact, all in the first row of 16 lanes): the increment of
 * the contact whose turn it is reaches everybody's residuals through the DPP operand of the FMA itself
 * (ROWBC_FMAC: v_fmac_f64_dpp row_newbcast), so the dependent path of a normal update is
 *   fmac (residual) -> fma (candidate) -> cmp -> select -> sub (increment) -> fmac ...
 * with no scalar-register round trip.  The contact loop is unrolled in blocks of RKFD_PGS_BLK (the broadcast lane is
 * a literal); a block's matrix entries are fetched from LDS together at its top, off the dependent path, instead
 * of three / six loads in front of every update.  Same arithmetic and update order as the general loop in
 * rkfd_phase_mlcp (reference src/rkfd_mlcp.c:190-249). */
#define RKFD_PGS_BLK 4
#define RKFD_PGS_DPP_MAX 16
template<bool pk, int C0> RKFD_DEV void rkfd_pgs_dpp_normal(const double *MA, int r0, int ld, int nc, int lane, double in_,
                                                            double &rn, double &r1, double &r2, double &fn)
{
  double a0[RKFD_PGS_BLK], a1[RKFD_PGS_BLK], a2[RKFD_PGS_BLK];
#pragma unroll
  for( int u=0; u<RKFD_PGS_BLK; u++ ){
    const int c = C0+u < nc ? C0+u : nc-1;       /* (a column beyond the last contact is not used: stay inside the matrix) */
    a0[u] = MA[rkfd_ma_idx_t<pk>( r0, 3*c, ld )]; a1[u] = MA[rkfd_ma_idx_t<pk>( r0+1, 3*c, ld )]; a2[u] = MA[rkfd_ma_idx_t<pk>( r0+2, 3*c, ld )];
  }
#define RKFD_PGS_N(u) \
  if( C0+u < nc ){ \
    /* normal force of contact c: f_n <- max( 0, -( b + a.f - a_nn f_n ) / a_nn ) */ \
    double ff = fn - rn*in_; \
    if( ff < RKFD_DEV_TOL ) ff = 0.0; \
    const double dl = ff - fn; \
    if( lane == C0+u ) fn = ff; \
    ROWBC_FMAC( C0+u, rn, dl, a0[u] ); ROWBC_FMAC( C0+u, r1, dl, a1[u] ); ROWBC_FMAC( C0+u, r2, dl, a2[u] ); \
  }
  RKFD_PGS_N(0) RKFD_PGS_N(1) RKFD_PGS_N(2) RKFD_PGS_N(3)
#undef RKFD_PGS_N
}
template<bool pk, int C0> RKFD_DEV void rkfd_pgs_dpp_tangent(const double *MA, int r0, int ld, int nc, int lane, double i1, double i2, double fs,
                                                             double &rn, double &r1, double &r2, double &f1, double &f2)
{
  /* (blocks of two: twelve matrix entries in flight, like the four contacts of a normal block) */
  double a0[2], a1[2], a2[2], b0[2], b1[2], b2[2];
#pragma unroll
  for( int u=0; u<2; u++ ){
    const int c = C0+u < nc ? C0+u : nc-1;
    a0[u] = MA[rkfd_ma_idx_t<pk>( r0, 3*c+1, ld )]; a1[u] = MA[rkfd_ma_idx_t<pk>( r0+1, 3*c+1, ld )]; a2[u] = MA[rkfd_ma_idx_t<pk>( r0+2, 3*c+1, ld )];
    b0[u] = MA[rkfd_ma_idx_t<pk>( r0, 3*c+2, ld )]; b1[u] = MA[rkfd_ma_idx_t<pk>( r0+1, 3*c+2, ld )]; b2[u] = MA[rkfd_ma_idx_t<pk>( r0+2, 3*c+2, ld )];
  }
#define RKFD_PGS_T(u) \
  if( C0+u < nc ){ \
    /* tangential forces of contact c: Gauss-Seidel value for both, then scaled onto the friction disc of radius mu f_n \
     * (fs = ( mu f_n )^2: the normal forces do not change during the tangential pass) */ \
    const double ff0 = f1 - r1*i1, ff1 = f2 - r2*i2; \
    const double fnorm = ff0*ff0 + ff1*ff1; \
    const bool zero = fnorm < RKFD_DEV_TOL || fs < RKFD_DEV_TOL; \
    double n1 = zero ? 0.0 : ff0, n2 = zero ? 0.0 : ff1; \
    /* only the decision of lane c matters: branch on it wave-uniformly, so that the reciprocal is evaluated only \
     * when contact c really slides */ \
    const bool sl = !zero && fnorm > fs; \
    if( ANY( sl && lane == C0+u ) ){ \
      const double sc = fs*RKFD_RCP( fnorm ); \
      n1 = sl ? ff0*sc : n1; n2 = sl ? ff1*sc : n2; \
    } \
    const double d1 = n1 - f1, d2 = n2 - f2; \
    if( lane == C0+u ){ f1 = n1; f2 = n2; } \
    ROWBC_FMAC( C0+u, rn, d2, b0[u] ); ROWBC_FMAC( C0+u, r1, d2, b1[u] ); ROWBC_FMAC( C0+u, r2, d2, b2[u] ); \
    ROWBC_FMAC( C0+u, rn, d1, a0[u] ); ROWBC_FMAC( C0+u, r1, d1, a1[u] ); ROWBC_FMAC( C0+u, r2, d1, a2[u] ); \
  }
  RKFD_PGS_T(0) RKFD_PGS_T(1)
#undef RKFD_PGS_T
}
template<bool pk> RKFD_DEV void rkfd_pgs_dpp(const double *MA, int r0, int ld, int nc, int maxrg, int max_iter, int lane, double mu,
                                             double in_, double i1, double i2, double &rn, double &r1, double &r2, double &fn, double &f1, double &f2)
{
  for( int it=0; it<max_iter; it++ ){
    rkfd_pgs_dpp_normal<pk, 0>( MA, r0, ld, nc, lane, in_, rn, r1, r2, fn );
    if( maxrg > 4 && nc > 4 ) rkfd_pgs_dpp_normal<pk, 4>( MA, r0, ld, nc, lane, in_, rn, r1, r2, fn );
    if( maxrg > 8 && nc > 8 ) rkfd_pgs_dpp_normal<pk, 8>( MA, r0, ld, nc, lane, in_, rn, r1, r2, fn );
    if( maxrg > 12 && nc > 12 ) rkfd_pgs_dpp_normal<pk, 12>( MA, r0, ld, nc, lane, in_, rn, r1, r2, fn );
    double fs = mu*fn; fs = fs*fs;
    rkfd_pgs_dpp_tangent<pk, 0>( MA, r0, ld, nc, lane, i1, i2, fs, rn, r1, r2, f1, f2 );
#define RKFD_PGS_TB(C0) if( maxrg > C0 && nc > C0 ) rkfd_pgs_dpp_tangent<pk, C0>( MA, r0, ld, nc, lane, i1, i2, fs, rn, r1, r2, f1, f2 );
    RKFD_PGS_TB(2) RKFD_PGS_TB(4) RKFD_PGS_TB(6) RKFD_PGS_TB(8) RKFD_PGS_TB(10) RKFD_PGS_TB(12) RKFD_PGS_TB(14)
#undef RKFD_PGS_TB
  }
}

/* GROUPED projected Gauss-Seidel (many contacts on several independent bodies: config 5's humanoid + four boxes).  Two contacts
 * are coupled only when they share a moving tree (the reference zeroes the other entries, src/rkfd_mlcp.c:76-102; here they are
 * exact zeros because the probe paths share no joint), so the contact problem falls into connected components, and an update
 * changes the residuals of its own component only.  The components are laid out one (or several, one after the other) per DPP
 * row of 16 lanes: the SAME instruction stream then runs the Gauss-Seidel sequence of every row at once - row_newbcast:C
 * delivers the increment of position C of EACH row within that row - and a sweep is as long as the fullest row (config 5: 8 + 8
 * instead of 24 + 24 updates).  Within a component the reference's order is kept; updates of different components never
 * touch the same residual, so the result is bit for bit that of the one-after-the-other loop
 * (tests/test_gpu_sustained.py::test_grouped_gauss_seidel_is_bit_identical; m:pgs 862 k -> 182 k cycles per step with the sweep-order
 * storage below, config 5 1.25 -> 2.77 M steps/s).  grow: this lane's row of the position table (contact index or 255), pos = lane & 15. */
/* position c (a literal) of a row of the position table held in four registers */
#define RKFD_GRP_POS(gw, c) ( (int)( ( (gw)[(c) >> 2] >> ( 8*( (c) & 3 ) ) ) & 255u ) )
/* entry ( row, col ) of the contact matrix where the row's own part of the packed index is known: rb = row ( row + 1 ) / 2 (per lane,
 * fixed for the solve), cbase = col ( col + 1 ) / 2 (from the position's contact) - no multiplication per entry */
template<bool pk> RKFD_DEV int rkfd_ma_idx2(int row, int rb, int col, int cbase, int ld)
{
  if( !pk ) return col*ld + row;      /* (the transposed position: see rkfd_ma_idx_t) */
  return col <= row ? rb + col : cbase + row;
}
template<bool pk, int C0> RKFD_DEV void rkfd_pgs_grp_normal(const double *MA, const unsigned *gw, int kfb, int r0, const int *rb, int ld, int maxlen, int pos, double in_,
                                                            double &rn, double &r1, double &r2, double &fn)
{
  /* (blocks of two: the entries of the second update are in flight while the first runs - measured 1.71 M against 1.62 M
   * steps/s on config 5 for one update at a time) */
  double a0[2], a1[2], a2[2];
#pragma unroll
  for( int u=0; u<2; u++ ){
    /* an empty position broadcasts a zero increment, but 0 x NaN is NaN: the entry must be one that was WRITTEN, and only the
     * blocks within a row are (kfb: this lane's own contact, or -1 on a lane without one, which takes entry ( 0, 0 ) whatever
     * the position holds and so stays finite, its own increments exact zeros) */
    int kc = RKFD_GRP_POS( gw, C0+u );
    if( kc == 255 ) kc = kfb;
    if( kfb < 0 ) kc = 0;
    const int c3 = 3*kc, cb = ( c3*( c3+1 ) ) >> 1;
    a0[u] = MA[rkfd_ma_idx2<pk>( r0, rb[0], c3, cb, ld )]; a1[u] = MA[rkfd_ma_idx2<pk>( r0+1, rb[1], c3, cb, ld )]; a2[u] = MA[rkfd_ma_idx2<pk>( r0+2, rb[2], c3, cb, ld )];
  }
#define RKFD_PGS_GN(u) \
  if( C0+u < maxlen ){ \
    double ff = fn - rn*in_; \
    if( ff < RKFD_DEV_TOL ) ff = 0.0; \
    const double dl = ff - fn; \
    if( pos == C0+u ) fn = ff; \
    ROWBC_FMAC( C0+u, rn, dl, a0[u] ); ROWBC_FMAC( C0+u, r1, dl, a1[u] ); ROWBC_FMAC( C0+u, r2, dl, a2[u] ); \
  }
  RKFD_PGS_GN(0) RKFD_PGS_GN(1)
#undef RKFD_PGS_GN
}
template<bool pk, int C0> RKFD_DEV void rkfd_pgs_grp_tangent(const double *MA, const unsigned *gw, int kfb, int r0, const int *rb, int ld, int maxlen, int pos, double i1, double i2, double fs,
                                                             double &rn, double &r1, double &r2, double &f1, double &f2)
{
  if( C0 < maxlen ){
    int kc = RKFD_GRP_POS( gw, C0 );
    if( kc == 255 ) kc = kfb;
    if( kfb < 0 ) kc = 0;
    const int c1 = 3*kc+1, c2 = c1+1, cb1 = ( c1*c2 ) >> 1, cb2 = cb1 + c2;      /* c ( c + 1 ) / 2 of the two tangential columns */
    const double a0 = MA[rkfd_ma_idx2<pk>( r0, rb[0], c1, cb1, ld )], a1 = MA[rkfd_ma_idx2<pk>( r0+1, rb[1], c1, cb1, ld )], a2 = MA[rkfd_ma_idx2<pk>( r0+2, rb[2], c1, cb1, ld )];
    const double b0 = MA[rkfd_ma_idx2<pk>( r0, rb[0], c2, cb2, ld )], b1 = MA[rkfd_ma_idx2<pk>( r0+1, rb[1], c2, cb2, ld )], b2 = MA[rkfd_ma_idx2<pk>( r0+2, rb[2], c2, cb2, ld )];
    const double ff0 = f1 - r1*i1, ff1 = f2 - r2*i2;
    const double fnorm = ff0*ff0 + ff1*ff1;
    const bool zero = fnorm < RKFD_DEV_TOL || fs < RKFD_DEV_TOL;
    double n1 = zero ? 0.0 : ff0, n2 = zero ? 0.0 : ff1;
    /* the lanes at position C0 of the rows decide for themselves; the branch stays wave-uniform (is any of them sliding?) and the
     * lanes select: a per-lane branch would rewrite EXEC right in front of the DPP instructions below, which sit in inline asm where
     * the compiler does not see that they need their wait states after an EXEC write */
    const bool sl = !zero && fnorm > fs;
    if( ANY( sl && pos == C0 ) ){
      const double sc = fs*RKFD_RCP( fnorm );
      n1 = sl ? ff0*sc : n1; n2 = sl ? ff1*sc : n2;
    }
    const double d1 = n1 - f1, d2 = n2 - f2;
    if( pos == C0 ){ f1 = n1; f2 = n2; }
    ROWBC_FMAC( C0, rn, d2, b0 ); ROWBC_FMAC( C0, r1, d2, b1 ); ROWBC_FMAC( C0, r2, d2, b2 );
    ROWBC_FMAC( C0, rn, d1, a0 ); ROWBC_FMAC( C0, r1, d1, a1 ); ROWBC_FMAC( C0, r2, d1, a2 );
  }
}
/* the layout: components -> rows; fills the position table tab (64 bytes of LDS: contact index or 255 per lane position) and
 * returns the four row fills packed in bytes, or -1 when the contacts do not fit (a component of more than 16 contacts, rows too
 * full).  Needs the packed moving-side records L.tgt. */
RKFD_DEV int rkfd_pgs_group_layout(const rkfdDevModel &m, const rkfdLds &L, unsigned char *tab, int nc)
{
  const int lane = LANE();
  const int NSD = m.nside;
  const unsigned long long below = lane == 0 ? 0ull : ( ~0ull >> ( 64-lane ) );
  /* the moving trees of this lane's contact (255: none) */
  int t0 = 255, t1 = 255;
  if( lane < nc ){
    const unsigned e0 = (unsigned)L.tgt[lane*NSD];
    if( RKFD_CS_VALID( e0 ) ) t0 = RKFD_CS_TOP( e0 );
    if( NSD > 1 ){ const unsigned e1 = (unsigned)L.tgt[lane*NSD+1]; if( RKFD_CS_VALID( e1 ) ) t1 = RKFD_CS_TOP( e1 ); }
  }
  /* the same contacts on the same trees as in the last evaluation (most evaluations: a step has five, the contact set changes
   * rarely): its layout again (36 k -> 7 k cycles per step on config 5) */
  unsigned short *gct = (unsigned short *)L.GC;
  const unsigned short mine = (unsigned short)( t0 | ( t1 << 8 ) );
  if( L.GC[RKFD_GC_INTS-1] == nc && BALLOT( gct[lane] != mine ) == 0ull ){
    if( lane < 16 ) ( (int *)tab )[lane] = L.GC[32+lane];
    SYNC();
    return L.GC[RKFD_GC_INTS-2];
  }
  /* the moving trees of this lane's contact as a bit mask over the links (a tree is named by its top link, < 64) */
  const unsigned long long trees = ( t0 != 255 ? 1ull << t0 : 0ull ) | ( t1 != 255 ? 1ull << t1 : 0ull );
  const int trlo = (int)( trees & 0xffffffffull ), trhi = (int)( trees >> 32 );
  /* connected components, in the order of their first contacts: grow a set of trees T from the first contact left - the contacts
   * touching T (one ballot), their trees (v_readlane, every contact once) - until it stops growing; every component goes to the
   * row that is emptiest so far: a sweep is as long as the fullest row */
  unsigned long long remaining = nc >= 64 ? ~0ull : ( ( 1ull << nc ) - 1ull );
  int f0 = 0, f1_ = 0, f2_ = 0, f3 = 0, maxlen = 0, target = -1;
  bool fits = true;
  /* lane i keeps component i (every component holds a contact: there are at most 64) */
  unsigned long long mycomp = 0ull;
  int ncomp = 0;
  while( remaining ){
    const int seed = __builtin_ctzll( remaining );
    unsigned long long T = ( (unsigned long long)(unsigned)BCASTI( trhi, seed ) << 32 ) | (unsigned)BCASTI( trlo, seed );
    unsigned long long comp = 1ull << seed, seen = comp;
    for(;;){
      const unsigned long long members = BALLOT( lane < nc && ( trees & T ) != 0ull ) & remaining;
      unsigned long long fresh = members & ~seen;
      comp |= members; seen |= members;
      if( !fresh ) break;
      while( fresh ){
        const int j = __builtin_ctzll( fresh );
        fresh &= fresh - 1ull;
        T |= ( (unsigned long long)(unsigned)BCASTI( trhi, j ) << 32 ) | (unsigned)BCASTI( trlo, j );
      }
    }
    remaining &= ~comp;
    if( lane == ncomp ) mycomp = comp;
    ncomp++;
  }
  /* rows: the largest components first (equal ones in the order found), each to the row that is emptiest so far - a sweep is as
   * long as the fullest row (config 5 standing: the boxes' contacts come before the humanoid's in the candidate order; first
   * come first served made rows of 12 4 4 4, largest first makes 8 8 4 4) */
  const int mysize = __builtin_popcountll( mycomp );
  const int mclo = (int)( mycomp & 0xffffffffull ), mchi = (int)( mycomp >> 32 );
  if( BALLOT( mysize > 16 ) != 0ull ) fits = false;
  for( int sz=16; sz>=1 && fits; sz-- ){
    unsigned long long todo = BALLOT( mysize == sz );
    while( todo ){
      const int j = __builtin_ctzll( todo );
      todo &= todo - 1ull;
      const unsigned long long comp = ( (unsigned long long)(unsigned)BCASTI( mchi, j ) << 32 ) | (unsigned)BCASTI( mclo, j );
      int row = 0, fill = f0;
      if( f1_ < fill ){ row = 1; fill = f1_; }
      if( f2_ < fill ){ row = 2; fill = f2_; }
      if( f3 < fill ){ row = 3; fill = f3; }
      if( fill + sz > 16 ){ fits = false; break; }
      if( ( comp >> lane ) & 1ull ) target = 16*row + fill + __builtin_popcountll( comp & below );
      fill += sz;
      if( row == 0 ) f0 = fill; else if( row == 1 ) f1_ = fill; else if( row == 2 ) f2_ = fill; else f3 = fill;
      if( fill > maxlen ) maxlen = fill;
    }
  }
  (void)maxlen;
  const int fills = fits ? ( f0 | ( f1_ << 8 ) | ( f2_ << 16 ) | ( f3 << 24 ) ) : -1;
  SYNC();       /* (everybody has compared with the remembered trees) */
  gct[lane] = mine;
  if( lane == 0 ){ L.GC[RKFD_GC_INTS-2] = fills; L.GC[RKFD_GC_INTS-1] = nc; }
  if( !fits ){ SYNC(); return -1; }
  tab[lane] = 255;
  SYNC();
  if( target >= 0 ) tab[target] = (unsigned char)lane;
  SYNC();
  if( lane < 16 ) L.GC[32+lane] = ( (const int *)tab )[lane];
  SYNC();
  return fills;
}
/* the solve on a layout made by rkfd_pgs_group_layout.  Writes MF itself. */
template<bool pk> RKFD_DEV void rkfd_pgs_grouped(const rkfdDevModel &m, const rkfdLds &L, const unsigned char *tab, int fills, int ld, double dt)
{
  const int lane = LANE();
  int maxlen = fills & 255;
  if( ( ( fills >> 8 ) & 255 ) > maxlen ) maxlen = ( fills >> 8 ) & 255;
  if( ( ( fills >> 16 ) & 255 ) > maxlen ) maxlen = ( fills >> 16 ) & 255;
  if( ( ( fills >> 24 ) & 255 ) > maxlen ) maxlen = ( fills >> 24 ) & 255;
  /* this lane's contact in the new layout */
  const int k = tab[lane];
  const bool on = k != 255;
  const int r0 = on ? 3*k : 0, pos = lane & 15, kfb = on ? k : -1;
  unsigned gw[4];        /* this lane's row of the table: 16 positions, one byte each */
  {
    const unsigned *g32 = (const unsigned *)&tab[lane & 48];
    gw[0] = g32[0]; gw[1] = g32[1]; gw[2] = g32[2]; gw[3] = g32[3];
  }
  const int rb[3] = { ( r0*( r0+1 ) ) >> 1, ( ( r0+1 )*( r0+2 ) ) >> 1, ( ( r0+2 )*( r0+3 ) ) >> 1 };
  double rn = 0, r1 = 0, r2 = 0, fn = 0, f1 = 0, f2 = 0, in_ = 0, i1 = 0, i2 = 0, mu = 0;
  if( on ){
    rn = L.MB[r0]; r1 = L.MB[r0+1]; r2 = L.MB[r0+2];
    const double dn = L.MA[rkfd_ma_idx<pk>( r0, r0, ld )], d1 = L.MA[rkfd_ma_idx<pk>( r0+1, r0+1, ld )], d2 = L.MA[rkfd_ma_idx<pk>( r0+2, r0+2, ld )];
    in_ = 1.0/dn;
    i1 = fabs( d1 ) < RKFD_DEV_TOL ? 0.0 : 1.0/d1;
    i2 = fabs( d2 ) < RKFD_DEV_TOL ? 0.0 : 1.0/d2;
    const int jr_ = L.lrg[k], cir_ = RKFD_CI_CI( L.CIp[jr_] );
    mu = L.typ[jr_] == RKFD_SF ? m.ci_sf[cir_] : m.ci_kf[cir_];
  }
  SYNC();       /* (MF shares its storage with MB in these kernels: everybody has read b before anybody writes f) */
  for( int it=0; it<m.max_iter; it++ ){
#ifndef RKFD_EMU
    /* (the table is re-read through an opaque asm every sweep: otherwise the compiler hoists the index arithmetic of all 48
     * updates out of the sweep loop and spills hundreds of registers) */
    asm volatile( "" : "+v"(gw[0]), "+v"(gw[1]), "+v"(gw[2]), "+v"(gw[3]) );
#endif
#define RKFD_PGS_GNB(C0) if( maxlen > C0 ) rkfd_pgs_grp_normal<pk, C0>( L.MA, gw, kfb, r0, rb, ld, maxlen, pos, in_, rn, r1, r2, fn );
    RKFD_PGS_GNB(0) RKFD_PGS_GNB(2) RKFD_PGS_GNB(4) RKFD_PGS_GNB(6) RKFD_PGS_GNB(8) RKFD_PGS_GNB(10) RKFD_PGS_GNB(12) RKFD_PGS_GNB(14)
#undef RKFD_PGS_GNB
    double fs = mu*fn; fs = fs*fs;
#define RKFD_PGS_GTB(C0) rkfd_pgs_grp_tangent<pk, C0>( L.MA, gw, kfb, r0, rb, ld, maxlen, pos, i1, i2, fs, rn, r1, r2, f1, f2 );
    RKFD_PGS_GTB(0) RKFD_PGS_GTB(1) RKFD_PGS_GTB(2) RKFD_PGS_GTB(3) RKFD_PGS_GTB(4) RKFD_PGS_GTB(5) RKFD_PGS_GTB(6) RKFD_PGS_GTB(7)
    RKFD_PGS_GTB(8) RKFD_PGS_GTB(9) RKFD_PGS_GTB(10) RKFD_PGS_GTB(11) RKFD_PGS_GTB(12) RKFD_PGS_GTB(13) RKFD_PGS_GTB(14) RKFD_PGS_GTB(15)
#undef RKFD_PGS_GTB
  }
  if( on ){ L.MF[r0] = fn/dt; L.MF[r0+1] = f1/dt; L.MF[r0+2] = f2/dt; }
}

/* SWEEP-ORDER storage of the blocks inside the rows (rows of at most 8 contacts: config 5's 8 + 8 + 8).  The grouped solve above
 * spends more instructions on WHERE an entry of the packed triangle lives (50 integer VALU operations per update, from the
 * position table to the triangle index) than on the update itself (17 fp64 operations) - and at one wave per SIMD (53 KB of LDS
 * per instance) nothing hides them.  When the fullest row has at most 8 contacts the matrix build stores, instead of the
 * triangle, for every lane of the new layout and every position c of its row the nine entries ( rows of the lane's contact ) x
 * ( columns of the contact at position c ) at  SW[ ( 9 c + 3 j + i ) * 32 + slot ],  slot = 8 * row + position of the lane:
 * the solve then reads with ONE address register per lane and literal offsets, no table, no index arithmetic; lanes of one
 * read are 8 bytes apart.  8 * 9 * 32 doubles = 18 KB, less than the packed triangle of 72 rows (21 KB).  Positions a row does
 * not fill hold zeros (the whole array is cleared before the build): their zero increments meet 0, not storage nobody wrote. */
#define RKFD_SW_SLOTS 32
#define RKFD_SW_MAXLEN 8
#define RKFD_SW_DOUBLES ( RKFD_SW_MAXLEN*9*RKFD_SW_SLOTS )
#define RKFD_SW_AT(c, j, i) ( ( 9*(c) + 3*(j) + (i) )*RKFD_SW_SLOTS )
template<int C0> RKFD_DEV void rkfd_pgs_sw_normal(const double *SWl, int maxlen, int pos, double in_, double &rn, double &r1, double &r2, double &fn)
{
  double a0[2], a1[2], a2[2];
#pragma unroll
  for( int u=0; u<2; u++ ){ a0[u] = SWl[RKFD_SW_AT( C0+u, 0, 0 )]; a1[u] = SWl[RKFD_SW_AT( C0+u, 0, 1 )]; a2[u] = SWl[RKFD_SW_AT( C0+u, 0, 2 )]; }
#define RKFD_PGS_GN(u) \
  if( C0+u < maxlen ){ \
    double ff = fn - rn*in_; \
    if( ff < RKFD_DEV_TOL ) ff = 0.0; \
    const double dl = ff - fn; \
    if( pos == C0+u ) fn = ff; \
    ROWBC_FMAC( C0+u, rn, dl, a0[u] ); ROWBC_FMAC( C0+u, r1, dl, a1[u] ); ROWBC_FMAC( C0+u, r2, dl, a2[u] ); \
  }
  RKFD_PGS_GN(0) RKFD_PGS_GN(1)
#undef RKFD_PGS_GN
}
template<int C0> RKFD_DEV void rkfd_pgs_sw_tangent(const double *SWl, int maxlen, int pos, double i1, double i2, double fs,
                                                   double &rn, double &r1, double &r2, double &f1, double &f2)
{
  if( C0 < maxlen ){
    const double a0 = SWl[RKFD_SW_AT( C0, 1, 0 )], a1 = SWl[RKFD_SW_AT( C0, 1, 1 )], a2 = SWl[RKFD_SW_AT( C0, 1, 2 )];
    const double b0 = SWl[RKFD_SW_AT( C0, 2, 0 )], b1 = SWl[RKFD_SW_AT( C0, 2, 1 )], b2 = SWl[RKFD_SW_AT( C0, 2, 2 )];
    const double ff0 = f1 - r1*i1, ff1 = f2 - r2*i2;
    const double fnorm = ff0*ff0 + ff1*ff1;
    const bool zero = fnorm < RKFD_DEV_TOL || fs < RKFD_DEV_TOL;
    double n1 = zero ? 0.0 : ff0, n2 = zero ? 0.0 : ff1;
    const bool sl = !zero && fnorm > fs;       /* (wave-uniform branch, lanes select: see rkfd_pgs_grp_tangent) */
    if( ANY( sl && pos == C0 ) ){
      const double sc = fs*RKFD_RCP( fnorm );
      n1 = sl ? ff0*sc : n1; n2 = sl ? ff1*sc : n2;
    }
    const double d1 = n1 - f1, d2 = n2 - f2;
    if( pos == C0 ){ f1 = n1; f2 = n2; }
    ROWBC_FMAC( C0, rn, d2, b0 ); ROWBC_FMAC( C0, r1, d2, b1 ); ROWBC_FMAC( C0, r2, d2, b2 );
    ROWBC_FMAC( C0, rn, d1, a0 ); ROWBC_FMAC( C0, r1, d1, a1 ); ROWBC_FMAC( C0, r2, d1, a2 );
  }
}
/* the solve on the sweep-order storage (L.MA holds SW).  Same updates in the same order as rkfd_pgs_grouped.  Writes MF itself. */
RKFD_DEV void rkfd_pgs_grouped_sw(const rkfdDevModel &m, const rkfdLds &L, const unsigned char *tab, int maxlen, double dt)
{
  const int lane = LANE();
  const int k = tab[lane];
  const bool on = k != 255;
  const int r0 = on ? 3*k : 0, pos = lane & 15;
  /* (a lane without a contact beyond position 7 has no slot: it reads slot 0, finite numbers, and its own increments stay exact
   * zeros because its inverse diagonals are zero) */
  int soff = pos < RKFD_SW_MAXLEN ? RKFD_SW_MAXLEN*( lane >> 4 ) + pos : 0;
  double rn = 0, r1 = 0, r2 = 0, fn = 0, f1 = 0, f2 = 0, in_ = 0, i1 = 0, i2 = 0, mu = 0;
  if( on ){
    rn = L.MB[r0]; r1 = L.MB[r0+1]; r2 = L.MB[r0+2];
    const double *dg = L.MA + soff + 9*pos*RKFD_SW_SLOTS;
    const double dn = dg[RKFD_SW_AT( 0, 0, 0 )], d1 = dg[RKFD_SW_AT( 0, 1, 1 )], d2 = dg[RKFD_SW_AT( 0, 2, 2 )];
    in_ = 1.0/dn;
    i1 = fabs( d1 ) < RKFD_DEV_TOL ? 0.0 : 1.0/d1;
    i2 = fabs( d2 ) < RKFD_DEV_TOL ? 0.0 : 1.0/d2;
    const int jr_ = L.lrg[k], cir_ = RKFD_CI_CI( L.CIp[jr_] );
    mu = L.typ[jr_] == RKFD_SF ? m.ci_sf[cir_] : m.ci_kf[cir_];
  }
  SYNC();       /* (MF shares its storage with MB in these kernels: everybody has read b before anybody writes f) */
  for( int it=0; it<m.max_iter; it++ ){
#ifndef RKFD_EMU
    /* (the address is re-read through an opaque asm every sweep: the entries do not change between sweeps, and the compiler
     * would otherwise keep all 72 of them in registers it does not have) */
    asm volatile( "" : "+v"(soff) );
#endif
    const double *SWl = L.MA + soff;
    rkfd_pgs_sw_normal<0>( SWl, maxlen, pos, in_, rn, r1, r2, fn );
    if( maxlen > 2 ) rkfd_pgs_sw_normal<2>( SWl, maxlen, pos, in_, rn, r1, r2, fn );
    if( maxlen > 4 ) rkfd_pgs_sw_normal<4>( SWl, maxlen, pos, in_, rn, r1, r2, fn );
    if( maxlen > 6 ) rkfd_pgs_sw_normal<6>( SWl, maxlen, pos, in_, rn, r1, r2, fn );
    double fs = mu*fn; fs = fs*fs;
#define RKFD_PGS_GTB(C0) rkfd_pgs_sw_tangent<C0>( SWl, maxlen, pos, i1, i2, fs, rn, r1, r2, f1, f2 );
    RKFD_PGS_GTB(0) RKFD_PGS_GTB(1) RKFD_PGS_GTB(2) RKFD_PGS_GTB(3) RKFD_PGS_GTB(4) RKFD_PGS_GTB(5) RKFD_PGS_GTB(6) RKFD_PGS_GTB(7)
#undef RKFD_PGS_GTB
  }
  if( on ){ L.MF[r0] = fn/dt; L.MF[r0+1] = f1/dt; L.MF[r0+2] = f2/dt; }
}

/* the same for exactly 8 contacts - the humanoid standing on both soles, the contact problem the headline workload
 * solves in almost every evaluation: the contact count is a literal, so the per-update guards of the block form above
 * (a scalar compare, a branch and the compiler's mask bookkeeping per update) are gone.  Same arithmetic and order.
 * Measured in isolation (tools/ubench/pgs.hip, eleven waves per CU): 265 cycles per update against 312 for the
 * guarded blocks and 368 for the general loop. */
template<bool pk> RKFD_DEV void rkfd_pgs_dpp8(const double *MA, int r0, int ld, int max_iter, int lane, double mu,
                                              double in_, double i1, double i2, double &rn, double &r1, double &r2, double &fn, double &f1, double &f2)
{
  for( int it=0; it<max_iter; it++ ){
#ifndef RKFD_EMU
    /* (the row index is re-read through an opaque asm every sweep: the entries do not change between sweeps, a lane's three
     * are neighbours in memory and their offsets literals - the compiler would otherwise load all 216 of them once, in front
     * of the sweeps, into registers it does not have: 425 spills) */
    asm volatile( "" : "+v"(r0) );
#endif
#define RKFD_PGS_N8(C0) { \
      double a0[4], a1[4], a2[4]; \
      _Pragma("unroll") for( int u=0; u<4; u++ ){ \
        a0[u] = MA[rkfd_ma_idx_t<pk>( r0, 3*( C0+u ), ld )]; a1[u] = MA[rkfd_ma_idx_t<pk>( r0+1, 3*( C0+u ), ld )]; a2[u] = MA[rkfd_ma_idx_t<pk>( r0+2, 3*( C0+u ), ld )]; } \
      RKFD_PGS_N1( C0, 0 ) RKFD_PGS_N1( C0, 1 ) RKFD_PGS_N1( C0, 2 ) RKFD_PGS_N1( C0, 3 ) }
#define RKFD_PGS_N1(C0, u) { \
      double ff = fn - rn*in_; \
      if( ff < RKFD_DEV_TOL ) ff = 0.0; \
      const double dl = ff - fn; \
      if( lane == C0+u ) fn = ff; \
      ROWBC_FMAC( C0+u, rn, dl, a0[u] ); ROWBC_FMAC( C0+u, r1, dl, a1[u] ); ROWBC_FMAC( C0+u, r2, dl, a2[u] ); }
    RKFD_PGS_N8( 0 ) RKFD_PGS_N8( 4 )
#undef RKFD_PGS_N1
#undef RKFD_PGS_N8
    double fs = mu*fn; fs = fs*fs;
#define RKFD_PGS_T1(c) { \
      const double a0 = MA[rkfd_ma_idx_t<pk>( r0, 3*c+1, ld )], a1 = MA[rkfd_ma_idx_t<pk>( r0+1, 3*c+1, ld )], a2 = MA[rkfd_ma_idx_t<pk>( r0+2, 3*c+1, ld )]; \
      const double b0 = MA[rkfd_ma_idx_t<pk>( r0, 3*c+2, ld )], b1 = MA[rkfd_ma_idx_t<pk>( r0+1, 3*c+2, ld )], b2 = MA[rkfd_ma_idx_t<pk>( r0+2, 3*c+2, ld )]; \
      const double ff0 = f1 - r1*i1, ff1 = f2 - r2*i2; \
      const double fnorm = ff0*ff0 + ff1*ff1; \
      const bool zero = fnorm < RKFD_DEV_TOL || fs < RKFD_DEV_TOL; \
      double n1 = zero ? 0.0 : ff0, n2 = zero ? 0.0 : ff1; \
      const bool sl = !zero && fnorm > fs; \
      if( ANY( sl && lane == c ) ){ \
        const double sc = fs*RKFD_RCP( fnorm ); \
        n1 = sl ? ff0*sc : n1; n2 = sl ? ff1*sc : n2; \
      } \
      const double d1 = n1 - f1, d2 = n2 - f2; \
      if( lane == c ){ f1 = n1; f2 = n2; } \
      ROWBC_FMAC( c, rn, d2, b0 ); ROWBC_FMAC( c, r1, d2, b1 ); ROWBC_FMAC( c, r2, d2, b2 ); \
      ROWBC_FMAC( c, rn, d1, a0 ); ROWBC_FMAC( c, r1, d1, a1 ); ROWBC_FMAC( c, r2, d1, a2 ); }
    RKFD_PGS_T1( 0 ) RKFD_PGS_T1( 1 ) RKFD_PGS_T1( 2 ) RKFD_PGS_T1( 3 ) RKFD_PGS_T1( 4 ) RKFD_PGS_T1( 5 ) RKFD_PGS_T1( 6 ) RKFD_PGS_T1( 7 )
#undef RKFD_PGS_T1
  }
}

/* the general loop: any number of contacts, increments broadcast with v_readlane */
template<bool pk> RKFD_DEV void rkfd_pgs_general(const double *MA, int r0, int ld, int nc, int max_iter, int lane, double mu,
                                                 double in_, double i1, double i2, double &rn, double &r1, double &r2, double &fn, double &f1, double &f2)
{
for( int it=0; it<max_iter; it++ ){
    for( int c=0; c<nc; c++ ){
      /* normal force of contact c: f_n <- max( 0, -( b + a.f - a_nn f_n ) / a_nn ) */
      const double a0 = MA[rkfd_ma_idx<pk>( r0, 3*c, ld )], a1 = MA[rkfd_ma_idx<pk>( r0+1, 3*c, ld )], a2 = MA[rkfd_ma_idx<pk>( r0+2, 3*c, ld )];
      double ff = fn - rn*in_;
      if( ff < RKFD_DEV_TOL ) ff = 0.0;
      const double dl = BCAST( ff - fn, c );
      if( lane == c ) fn = ff;
      rn = fma( a0, dl, rn ); r1 = fma( a1, dl, r1 ); r2 = fma( a2, dl, r2 );
    }
    for( int c=0; c<nc; c++ ){
      /* tangential forces of contact c: Gauss-Seidel value for both, then scaled onto the
       * friction disc of radius mu f_n */
      const double a0 = MA[rkfd_ma_idx<pk>( r0, 3*c+1, ld )], a1 = MA[rkfd_ma_idx<pk>( r0+1, 3*c+1, ld )], a2 = MA[rkfd_ma_idx<pk>( r0+2, 3*c+1, ld )];
      const double b0 = MA[rkfd_ma_idx<pk>( r0, 3*c+2, ld )], b1 = MA[rkfd_ma_idx<pk>( r0+1, 3*c+2, ld )], b2 = MA[rkfd_ma_idx<pk>( r0+2, 3*c+2, ld )];
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
}

/* (The contact-space matrix was also built as a Gram product A = N'N on the matrix cores - v_mfma_f64_16x16x4_f64, K = the joints -
 * in round 2, measured against the block loops below, and found slower: 25.1 k -> 61.1 k cycles per instance-step on config 4, the
 * padded 32 x 32 x 31 product does 37 k FMAs where 2.9 k are useful and building the masked operands costs as many VALU
 * instructions as the loops it replaces (profiles/r02_mfma_ab.txt, r02_mfma_counters.json).  The variant was removed in round 3 -
 * dead code in the product; git history keeps it: rkfd_mlcp_matrix_mfma.  The matrix cores do serve the Vert plugin's Q = A'A,
 * rkfd_dev_vertqp.h.) */

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
template<bool prof, bool vqp, bool pk> RKFD_DEV void rkfd_phase_mlcp(const rkfdDevModel &m, const rkfdLds &L, const double *bv, bool doUpRef, int dtask, unsigned long long *pc)
{
  /* the Vert plugin's rigid branch (reference src/rkfd_vert.c:325-336) shares the contact system (A, b) and
   * the way the forces are applied; it differs in the solver (QP instead of PGS), in where the
   * relaxation enters (the QP objective, not A) and in when contact state is committed */
  const bool vert = vqp;     /* (the host launches the QP-carrying variant exactly for the worlds with rigid pairs under the Vert plugin, dm.vert_rigid:
                              * the Gauss-Seidel code is not part of that kernel at all) */
  unsigned long long q0 = prof ? RKFD_CLOCK() : 0ull, q1;
#define MST(k) do{ if( prof ){ q1 = RKFD_CLOCK(); pc[k] += q1 - q0; q0 = q1; } }while(0)
  const int lane = LANE();
  const int nc = L.cnt[CNT_NRG];
  const int M = 3*nc;
  const int ld = ( m.vert_rigid || nc < m.maxrg ) ? M+1 : M;   /* odd row stride unless every slot is taken (see rkfd_lds_carve) */
  const int NLV = m.nlevel, NL = m.nlink, NR = m.npurow;
  const int NSD = m.nside;
  const int PUS = 3*m.maxrg;                          /* columns between the two sides of PU (RKFD_PU_AT) */
  const unsigned char *TOP = L.PL + NL*NLV;           /* where a force on a link stops propagating (255: static) */
  const unsigned char *FSL = TOP + NL;                /* float slot of a link */
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
    double axl[9];
    d_load_axes( L, L.asl[j], axl );
#pragma unroll
    for( int i=0; i<3; i++ ){
      const double *ax = &axl[3*i];
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
      const int d0 = RKFD_LI_DEPTH( lit ) + ( RKFD_JT_IS1( jtt ) ? 0 : 1 );
      const unsigned e = (unsigned)a | ( (unsigned)RKFD_LI_DEPTH( L.LI[a] ) << 8 ) | ( (unsigned)top << 14 ) | ( (unsigned)d0 << 22 )
                       | ( jtt == RKFD_JOINT_FLOAT ? 1u << 29 : 0u ) | ( (unsigned)sd << 30 ) | ( top != 255 ? 1u << 31 : 0u );
      L.tgt[NSD == 1 ? lane : 2*lane+sd] = (int)e;
    }
  }
  /* sqrt(1/D) of the 1-DoF joints (MS slot 2: the driving torque kept there is dead after sweep 2) */
  if( lane < NL ){
    const int jt = RKFD_LI_JT( L.LI[lane] );
    if( RKFD_JT_IS1( jt ) ) L.MS[3*lane+2] = sqrt( L.MS[3*lane+0] );
  }
  if( m.has_brf ) rkfd_brf_before_probes( m, L );
  SYNC();
  /* many contacts on several independent bodies: the grouped layout (rkfd_pgs_group_layout) serves the matrix build and the
   * Gauss-Seidel below; its table lives in the link accelerations' storage, free from here to the delta sweep */
  MST(14);
  int gfills = -1;
  if( !vert && m.maxrg > RKFD_PGS_DPP_MAX && nc > RKFD_PGS_DPP_MAX && !( m.mlcp_mfma & 8 ) )
    gfills = rkfd_pgs_group_layout( m, L, (unsigned char *)L.AC, nc );
  /* fullest row; rows of at most 8: the blocks go to the sweep-order storage (rkfd_pgs_grouped_sw) */
  int gmaxlen = 0;
  if( gfills >= 0 ){
    gmaxlen = gfills & 255;
    if( ( ( gfills >> 8 ) & 255 ) > gmaxlen ) gmaxlen = ( gfills >> 8 ) & 255;
    if( ( ( gfills >> 16 ) & 255 ) > gmaxlen ) gmaxlen = ( gfills >> 16 ) & 255;
    if( ( ( gfills >> 24 ) & 255 ) > gmaxlen ) gmaxlen = ( gfills >> 24 ) & 255;
  }
  const bool sw = gfills >= 0 && gmaxlen <= RKFD_SW_MAXLEN && m.ma_size >= RKFD_SW_DOUBLES && !( m.mlcp_mfma & 32 );
  MST(31);
  /* probes: lane = column k = 3c+i; unit force along axis i at contact c, applied to the
   * owner link (+) and the other link (-).  Every level between the contact link and the top of
   * its path carries a 1-DoF joint; the operands of the next level are fetched while this one is
   * computed. */
  for( int cb=0; cb<M; cb+=RKFD_WL ){      /* 64 probe columns at a time (32 with two instances per wavefront) */
    const int col = cb + lane;
    const bool on = col < M;
    const int c = on ? col/3 : 0, ia = on ? col%3 : 0;
    const int j = L.lrg[c];
    double W[6];
    {
      const double x[3] = { L.CX[3*L.asl[j]], L.CX[3*L.asl[j]+1], L.CX[3*L.asl[j]+2] };
      double axl[9], ax[3];
      d_load_axes( L, L.asl[j], axl );
#pragma unroll
      for( int k=0; k<3; k++ ) ax[k] = ia == 0 ? axl[k] : ( ia == 1 ? axl[3+k] : axl[6+k] );
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
        double *pu = &L.PU[RKFD_PU_AT( m, s2, col, 0 )];
        const unsigned char *path = &L.PL[a*NLV];
        double Sx[6], Ux[6], sdx, dix;
#pragma unroll
        for( int k=0; k<6; k++ ){ Sx[k] = L.S[6*a+k]; Ux[k] = L.U[6*a+k]; }
        sdx = L.MS[3*a+2]; dix = L.MS[3*a+0];
        int inext = path[da > 0 ? da-1 : 0];
        for( int d=da; d>=d0; d-- ){
          const int in_ = inext;
          double Sn[6], Un[6];
#pragma unroll
          for( int k=0; k<6; k++ ){ Sn[k] = L.S[6*in_+k]; Un[k] = L.U[6*in_+k]; }
          const double sdn = L.MS[3*in_+2], din = L.MS[3*in_+0];
          inext = path[d > 1 ? d-2 : 0];
          double du0 = Sx[0]*dp[0], du1 = Sx[1]*dp[1];
          du0 = fma( Sx[2], dp[2], du0 ); du1 = fma( Sx[3], dp[3], du1 );
          du0 = fma( Sx[4], dp[4], du0 ); du1 = fma( Sx[5], dp[5], du1 );
          const double du = -( du0 + du1 );
          pu[d] = du*sdx;
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
          double Lr[21];
          d_chol6_load( &L.CHOL[21*FSL[RKFD_CS_TOP( e )]], Lr );
          d_chol6_fwd( Lr, rhs, y );
#pragma unroll
          for( int k=0; k<6; k++ ) pu[NLV+k] = y[k];
        }
      }
    }
  }
  SYNC();
  MST(15);
  /* (grouped layout: only the blocks inside the rows are ever read - a row holds whole components, blocks between rows are the
   * exact zeros between independent bodies; config 5: 92 blocks in two passes instead of 300 in five) */
  const int gf0 = gfills & 255, gf1 = ( gfills >> 8 ) & 255, gf2 = ( gfills >> 16 ) & 255, gf3 = ( gfills >> 24 ) & 255;
  const int gb1 = ( gf0*( gf0+1 ) ) >> 1, gb2 = gb1 + ( ( gf1*( gf1+1 ) ) >> 1 ), gb3 = gb2 + ( ( gf2*( gf2+1 ) ) >> 1 ), gb4 = gb3 + ( ( gf3*( gf3+1 ) ) >> 1 );
  const int nblk = gfills >= 0 ? gb4 : ( pk ? ( nc*( nc+1 ) >> 1 ) : nc*nc );
  {
  if( sw ){
    for( int i=lane; i<9*RKFD_SW_SLOTS*gmaxlen; i+=RKFD_WL ) L.MA[i] = 0.0;
    SYNC();
  }
  /* A, one 3x3 block per lane and pass: block ( cr, ck <= cr ) and its mirror image */
  for( int e0=0; e0<nblk; e0+=RKFD_WL ){
    const int e = e0 + lane;
    int cr, ck; bool one;
    int swr = 0, swk = 0, posr = 0, posk = 0;      /* sweep-order storage: slots and positions of the two contacts */
    if( gfills >= 0 ){
      const unsigned char *tab = (const unsigned char *)L.AC;
      const int row = e < gb1 ? 0 : ( e < gb2 ? 1 : ( e < gb3 ? 2 : 3 ) );
      const int t = e - ( row == 0 ? 0 : ( row == 1 ? gb1 : ( row == 2 ? gb2 : gb3 ) ) );
      int a = (int)( ( sqrt( 8.0*t + 1.0 ) - 1.0 )*0.5 );
      if( ( a*( a+1 ) >> 1 ) > t ) a--;
      if( ( ( a+1 )*( a+2 ) >> 1 ) <= t ) a++;
      const int b = t - ( a*( a+1 ) >> 1 );
      one = e < gb4;
      const int ka = one ? tab[16*row+a] : 0, kb = one ? tab[16*row+b] : 0;
      cr = ka > kb ? ka : kb; ck = ka > kb ? kb : ka;
      posr = ka > kb ? a : b; posk = ka > kb ? b : a;
      swr = RKFD_SW_MAXLEN*row + posr; swk = RKFD_SW_MAXLEN*row + posk;
    } else if( pk ){
      /* (the packed-matrix kernels serve the worlds with many contacts: lane = block of the lower triangle counted row by row, so
       * that every lane of a pass has one - 24 contacts: 5 passes instead of 9) */
      cr = (int)( ( sqrt( 8.0*e + 1.0 ) - 1.0 )*0.5 );
      if( ( cr*( cr+1 ) >> 1 ) > e ) cr--;
      if( ( ( cr+1 )*( cr+2 ) >> 1 ) <= e ) cr++;
      ck = e - ( cr*( cr+1 ) >> 1 );
      one = e < ( nc*( nc+1 ) >> 1 );
    } else {
      cr = e/nc; ck = e - cr*nc;
      one = e < nc*nc && ck <= cr;
    }
    if( one ){
      double blk[9] = {0,0,0,0,0,0,0,0,0};
      for( int sr=0; sr<NSD; sr++ ) for( int sk=0; sk<NSD; sk++ ){
        const unsigned er = (unsigned)L.tgt[cr*NSD+sr], ek = (unsigned)L.tgt[ck*NSD+sk];
        if( !RKFD_CS_VALID( er ) || !RKFD_CS_VALID( ek ) || RKFD_CS_TOP( er ) != RKFD_CS_TOP( ek ) ) continue;   /* no joint in common */
        const double *pr = &L.PU[RKFD_PU_AT( m, sr, 3*cr, 0 )], *pkk = &L.PU[RKFD_PU_AT( m, sk, 3*ck, 0 )];
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
          const double r0 = pr[d], r1 = pr[NR+d], r2 = pr[2*NR+d];
          const double k0 = pkk[d], k1 = pkk[NR+d], k2 = pkk[2*NR+d];
          blk[0] = fma( r0, k0, blk[0] ); blk[1] = fma( r0, k1, blk[1] ); blk[2] = fma( r0, k2, blk[2] );
          blk[3] = fma( r1, k0, blk[3] ); blk[4] = fma( r1, k1, blk[4] ); blk[5] = fma( r1, k2, blk[5] );
          blk[6] = fma( r2, k0, blk[6] ); blk[7] = fma( r2, k1, blk[7] ); blk[8] = fma( r2, k2, blk[8] );
        }
        if( RKFD_CS_FLOAT( er ) ){
#pragma unroll
          for( int q=0; q<6; q++ ){
            const int d = NLV + q;
            const double r0 = pr[d], r1 = pr[NR+d], r2 = pr[2*NR+d];
            const double k0 = pkk[d], k1 = pkk[NR+d], k2 = pkk[2*NR+d];
            blk[0] = fma( r0, k0, blk[0] ); blk[1] = fma( r0, k1, blk[1] ); blk[2] = fma( r0, k2, blk[2] );
            blk[3] = fma( r1, k0, blk[3] ); blk[4] = fma( r1, k1, blk[4] ); blk[5] = fma( r1, k2, blk[5] );
            blk[6] = fma( r2, k0, blk[6] ); blk[7] = fma( r2, k1, blk[7] ); blk[8] = fma( r2, k2, blk[8] );
          }
        }
      }
      if( cr == ck && !vert ){
        /* relaxation on the diagonal */
        const double rl = m.ci_l[RKFD_CI_CI( L.CIp[L.lrg[cr]] )];
        blk[0] += rl; blk[4] += rl; blk[8] += rl;
      }
      if( sw ){
        /* entry ( i, q ) of the block belongs to the lane of cr at the position of ck, and - transposed - to the lane of ck at the
         * position of cr */
        double *wr = L.MA + 9*posk*RKFD_SW_SLOTS + swr, *wk = L.MA + 9*posr*RKFD_SW_SLOTS + swk;
#pragma unroll
        for( int i=0; i<3; i++ )
#pragma unroll
          for( int q=0; q<3; q++ ){
            wr[RKFD_SW_AT( 0, q, i )] = blk[3*i+q];
            if( cr != ck ) wk[RKFD_SW_AT( 0, i, q )] = blk[3*i+q];
          }
      } else
#pragma unroll
      for( int i=0; i<3; i++ )
#pragma unroll
        for( int q=0; q<3; q++ ){
          if( !pk ){
            L.MA[( 3*cr+i )*ld + 3*ck+q] = blk[3*i+q];
            if( cr != ck ) L.MA[( 3*ck+q )*ld + 3*cr+i] = blk[3*i+q];
          } else if( cr != ck || q <= i ){
            L.MA[rkfd_ma_idx<true>( 3*cr+i, 3*ck+q, 0 )] = blk[3*i+q];
          }
        }
    }
  }
  }
  SYNC();
  MST(6);
  if( vert ){
    /* (more unknowns or pyramid faces than lanes: the wide form, which leaves the active flags in L.QA) */
    const bool wide = m.vert_rigid == 3;
    unsigned long long qmask = 0;
    if( wide ) rkfd_vert_qp_wide<prof>( m, L, nc, pc ); else qmask = rkfd_vert_qp<prof>( m, L, nc, pc );
    MST(21);
    /* _rkFDSolverSetForce (reference src/rkfd_vert.c:286-323): contact state is committed only when doUpRef;
     * a vertex is in kinetic friction when one of its pyramid faces is active at the solution */
    if( lane < nc ){
      const int j = L.lrg[lane];
      double fw[3] = {0,0,0}, axl[9];
      d_load_axes( L, L.asl[j], axl );
#pragma unroll
      for( int i=0; i<3; i++ ){
        const double fi = L.MF[3*lane+i];
        fw[0] += fi*axl[3*i]; fw[1] += fi*axl[3*i+1]; fw[2] += fi*axl[3*i+2];
      }
      { const int sl_ = L.asl[j]; L.CF[3*sl_] = fw[0]; L.CF[3*sl_+1] = fw[1]; L.CF[3*sl_+2] = fw[2]; L.FS[sl_] = 1; }
      if( doUpRef ){
        const int P = m.pyramid;
        bool anyface = false;
        if( wide ){ for( int f=0; f<P; f++ ) anyface = anyface || L.QA[lane*P+f]; }
        else anyface = ( ( qmask >> ( lane*P ) ) & ( ( 1ull << P ) - 1ull ) ) != 0ull;
        if( anyface ){
          L.typ[j] = RKFD_KF;
          { const int ri = RIDX( j ), sl_ = L.asl[j]; L.REF[3*ri] = L.PRO[3*sl_]; L.REF[3*ri+1] = L.PRO[3*sl_+1]; L.REF[3*ri+2] = L.PRO[3*sl_+2]; }
        } else {
          L.typ[j] = RKFD_SF;
          if( m.has_slide ){ const int ri = RIDX( j ), sl_ = L.asl[j]; L.REF[3*ri] += L.SD[3*sl_]; L.REF[3*ri+1] += L.SD[3*sl_+1]; L.REF[3*ri+2] += L.SD[3*sl_+2]; }
        }
      }
    }
  } else {
  /* projected Gauss-Seidel, fixed max_iter sweeps, no warm start (_rkFDSolverMLCP, reference
   * src/rkfd_mlcp.c:190-249), same update order.  lane = contact: each lane keeps the three
   * residuals res = b + A f, forces and inverse diagonals of ITS contact in registers, every lane
   * evaluates its own Gauss-Seidel candidate, and only the increment of the contact whose turn it
   * is gets broadcast (v_readlane) and applied to everybody's residuals. */
  {
    const bool on = lane < nc;
    const int r0 = on ? 3*lane : 0;
    double rn = 0, r1 = 0, r2 = 0, fn = 0, f1 = 0, f2 = 0, in_ = 0, i1 = 0, i2 = 0, mu = 0;
    bool grouped = false;
    if( on ){
      rn = L.MB[r0]; r1 = L.MB[r0+1]; r2 = L.MB[r0+2];
      const double dn = L.MA[rkfd_ma_idx<pk>( r0, r0, ld )], d1 = L.MA[rkfd_ma_idx<pk>( r0+1, r0+1, ld )], d2 = L.MA[rkfd_ma_idx<pk>( r0+2, r0+2, ld )];
      in_ = 1.0/dn;
      /* tangential rows with |a_kk| < zTOL are frozen at 0 (reference :220-221) */
      i1 = fabs( d1 ) < RKFD_DEV_TOL ? 0.0 : 1.0/d1;
      i2 = fabs( d2 ) < RKFD_DEV_TOL ? 0.0 : 1.0/d2;
      const int jr_ = L.lrg[lane], cir_ = RKFD_CI_CI( L.CIp[jr_] );
      mu = L.typ[jr_] == RKFD_SF ? m.ci_sf[cir_] : m.ci_kf[cir_];
    }
    /* up to 4 contacts: the lane's matrix rows in registers; up to 16: increments broadcast through the DPP operand of
     * the FMA; more, or the packed matrix (worlds with more than 16 contacts; its index arithmetic in four-contact
     * blocks costs registers the kernel does not have): the general loop.  Measured in isolation on the box
     * (tools/ubench/pgs.hip, cycles per update, eleven waves per CU): 3 contacts 311 / 389 / 384 (registers / DPP /
     * general), 8 contacts 312 / 369 (DPP / general), 16 contacts 270 / 317.  Variants with fewer VALU instructions
     * (wave-uniform branches on the deciding lane instead of selects, one-lane moves under a narrowed EXEC) were
     * slower (8 contacts: 333): the VALU -> SALU -> branch round trips sit on the dependent path. */
#ifdef RKFD_TRIM_PGS
    if( gfills >= 0 && sw ){ rkfd_pgs_grouped_sw( m, L, (const unsigned char *)L.AC, gmaxlen, dt ); grouped = true; }
    else rkfd_pgs_general<pk>( L.MA, r0, ld, nc, m.max_iter, lane, mu, in_, i1, i2, rn, r1, r2, fn, f1, f2 );
#else
    if( nc <= RKFD_PGS_NC ) rkfd_pgs_registers<pk>( L.MA, r0, ld, nc, m.max_iter, on, lane, mu, in_, i1, i2, rn, r1, r2, fn, f1, f2 );
    else if( nc == 8 ) rkfd_pgs_dpp8<pk>( L.MA, r0, ld, m.max_iter, lane, mu, in_, i1, i2, rn, r1, r2, fn, f1, f2 );
    else if( ( !pk || RKFD_W == 2 ) && nc <= RKFD_PGS_DPP_MAX ) rkfd_pgs_dpp<pk>( L.MA, r0, ld, nc, m.maxrg, m.max_iter, lane, mu, in_, i1, i2, rn, r1, r2, fn, f1, f2 );
    else if( gfills >= 0 ){
      /* (several independent bodies in contact: their Gauss-Seidel sequences run side by side, one DPP row each) */
      if( sw ) rkfd_pgs_grouped_sw( m, L, (const unsigned char *)L.AC, gmaxlen, dt );
      else rkfd_pgs_grouped<pk>( m, L, (const unsigned char *)L.AC, gfills, ld, dt );
      grouped = true;
    }
    else rkfd_pgs_general<pk>( L.MA, r0, ld, nc, m.max_iter, lane, mu, in_, i1, i2, rn, r1, r2, fn, f1, f2 );
#endif
    if( on && !grouped ){ L.MF[r0] = fn/dt; L.MF[r0+1] = f1/dt; L.MF[r0+2] = f2/dt; }
  }
  SYNC();
  MST(21);
  /* _rkFDSolverSetForce (reference src/rkfd_mlcp.c:252-284) incl. quirks Q1 / Q2 */
  if( lane < nc ){
    const int j = L.lrg[lane], ci = RKFD_CI_CI( L.CIp[j] );
    double fw[3] = {0,0,0}, axl[9];
    d_load_axes( L, L.asl[j], axl );
#pragma unroll
    for( int i=0; i<3; i++ ){
      const double fi = L.MF[3*lane+i];
      fw[0] += fi*axl[3*i]; fw[1] += fi*axl[3*i+1]; fw[2] += fi*axl[3*i+2];
    }
    { const int sl_ = L.asl[j]; L.CF[3*sl_] = fw[0]; L.CF[3*sl_+1] = fw[1]; L.CF[3*sl_+2] = fw[2]; L.FS[sl_] = 1; }
    const double fn = fw[0], fs = sqrt( fw[1]*fw[1] + fw[2]*fw[2] );
    const double mu = L.typ[j] == RKFD_SF ? m.ci_sf[ci] : m.ci_kf[ci];
    if( fs > mu*fn - RKFD_DEV_TOL ){
      L.typ[j] = RKFD_KF;
      { const int ri = RIDX( j ), sl_ = L.asl[j]; L.REF[3*ri] = L.PRO[3*sl_]; L.REF[3*ri+1] = L.PRO[3*sl_+1]; L.REF[3*ri+2] = L.PRO[3*sl_+2]; }
    } else {
      L.typ[j] = RKFD_SF;
      if( m.has_slide ){ const int ri = RIDX( j ), sl_ = L.asl[j]; L.REF[3*ri] += L.SD[3*sl_]; L.REF[3*ri+1] += L.SD[3*sl_+1]; L.REF[3*ri+2] += L.SD[3*sl_+2]; }
    }
  }
  }
  SYNC();
  MST(22);
  /* inputs of the delta sweep, lane = joint coordinate (dtask: its link and component, fixed for the launch): what the solved
   * forces F = MF do to the joint's innovation.  1-DoF joint: sum_k F_k nu_k / D  (= scaled sum times sqrt(1/D));  float joint:
   * sum_k F_k y_k, six components.  Links no contact path passes get 0. */
  {
    const bool has = dtask >= 0;
    const int link = has ? dtask & 255 : 0, fq = has ? dtask >> 8 : 0;
    const int lii = L.LI[link], jt = RKFD_LI_JT( lii );
    const bool is1 = has && RKFD_JT_IS1( jt ), isf = has && !is1;
    const int dpt = is1 ? RKFD_LI_DEPTH( lii ) : m.pu_d0;
    const int row = isf ? NLV+fq : dpt;
    double sum = 0;
    /* one side of one contact: its share of this lane's sum (e: the side's record, wave-uniform) */
#define RKFD_DIN_ONP(e) ( isf ? RKFD_CS_TOP( e ) == link \
                             : ( RKFD_CS_DEPTH( e ) >= dpt && RKFD_CS_D0( e ) <= dpt && L.PL[RKFD_CS_LINK( e )*NLV+dpt] == link ) )
    /* (with two instances per wavefront these broadcasts go through the LDS crossbar - ds_bpermute - instead of a scalar register;
     *  still faster than the loop over LDS below: 11.03 against 10.52 M steps/s on config 4, profiles/r03_ipw_ab.txt) */
    if( nc*NSD <= RKFD_WL ){
      /* the records and the forces wait in registers (lane = side, lane = contact) and reach everybody through v_readlane: the
       * loop over the sides that move (the floor's do not: half of the sides where contacts may have two moving ones) then holds
       * no LDS access that depends on another, two sides are in flight together, and the sum is still taken in the order of the
       * sides */
      const int mytgt = lane < nc*NSD ? L.tgt[lane] : 0;
      double g0 = 0, g1 = 0, g2 = 0;
      if( lane < nc ){ g0 = L.MF[3*lane]; g1 = L.MF[3*lane+1]; g2 = L.MF[3*lane+2]; }
      unsigned long long todo = BALLOT( RKFD_CS_VALID( (unsigned)mytgt ) != 0 );
      const double *pur = &L.PU[RKFD_PU_AT( m, 0, 0, row )];
      while( todo ){
        const int csa = __builtin_ctzll( todo );
        todo &= todo - 1ull;
        const bool two = todo != 0ull;
        const int csb = two ? __builtin_ctzll( todo ) : csa;
        todo &= todo - 1ull;
        const unsigned ea = (unsigned)BCASTI( mytgt, csa ), eb = (unsigned)BCASTI( mytgt, csb );
        const int ca = NSD == 1 ? csa : csa >> 1, cb = NSD == 1 ? csb : csb >> 1;
        const double *pa = pur + ( ( NSD == 1 ? 0 : ( csa & 1 ) )*PUS + 3*ca )*NR, *pb = pur + ( ( NSD == 1 ? 0 : ( csb & 1 ) )*PUS + 3*cb )*NR;
        const double va = BCAST( g0, ca )*pa[0] + BCAST( g1, ca )*pa[NR] + BCAST( g2, ca )*pa[2*NR];
        const double vb = BCAST( g0, cb )*pb[0] + BCAST( g1, cb )*pb[NR] + BCAST( g2, cb )*pb[2*NR];
        /* (no short-circuit evaluation: as branches around the byte load the test serialised the two sides) */
        const int pla = L.PL[RKFD_CS_LINK( ea )*NLV+dpt], plb = L.PL[RKFD_CS_LINK( eb )*NLV+dpt];
        const bool o1a = ( RKFD_CS_DEPTH( ea ) >= dpt ) & ( RKFD_CS_D0( ea ) <= dpt ) & ( pla == link );
        const bool o1b = ( RKFD_CS_DEPTH( eb ) >= dpt ) & ( RKFD_CS_D0( eb ) <= dpt ) & ( plb == link );
        const bool ofa = RKFD_CS_TOP( ea ) == link, ofb = RKFD_CS_TOP( eb ) == link;
        const bool oa = isf ? ofa : o1a, ob = isf ? ofb : o1b;
        sum += oa ? va : 0.0;
        sum += ( two & ob ) ? vb : 0.0;
      }
    } else
    for( int cs=0; cs<nc*NSD; cs++ ){
      const unsigned e = (unsigned)BCASTI( L.tgt[cs], 0 );
      if( !RKFD_CS_VALID( e ) ) continue;
      const int c = NSD == 1 ? cs : cs >> 1;
      const double *pu = &L.PU[RKFD_PU_AT( m, NSD == 1 ? 0 : ( cs & 1 ), 3*c, row )];
      const double v = L.MF[3*c]*pu[0] + L.MF[3*c+1]*pu[NR] + L.MF[3*c+2]*pu[2*NR];
      /* 1-DoF joint: it lies on the moving path of the contact side; float joint: the path ends there */
      sum += RKFD_DIN_ONP( e ) ? v : 0.0;
    }
#undef RKFD_DIN_ONP
    if( is1 ) L.MS[3*link+1] = sum*L.MS[3*link+2];
    if( isf ) L.U[6*link+fq] = sum;
  }
  SYNC();
  MST(23);
#undef MST
}

#endif /* RKFD_DEV_MLCP_H */
