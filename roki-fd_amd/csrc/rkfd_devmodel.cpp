/* rkfd_devmodel.cpp - host-side construction of the device model tables from an rkfdModel.
 * Pure host C++ (no HIP): used by the C-ABI (rkfd_capi.hip), which copies the blob to HBM,
 * and by the lane emulator under tests/emu.
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <vector>
#include "rkfd_model.h"
#include "rkfd_devmodel.h"
#include "rkfd_devmodel_host.h"

namespace {
struct Blob {
  std::vector<char> buf;
  size_t put(const void *src, size_t bytes){
    size_t off = ( buf.size() + 15 ) & ~(size_t)15;
    buf.resize( off + ( bytes ? bytes : 8 ) );
    if( src && bytes ) memcpy( buf.data()+off, src, bytes );
    return off;
  }
};
}

extern "C" int rkfd_devmodel_build(const rkfdModel *m, int max_rigid, rkfdDevModelHost *out, char *err, int errlen)
{
  const int NL = m->nlink, ND = m->ndof, NC = m->ncand;
#define FAIL(...) do{ if( err ) snprintf( err, errlen, __VA_ARGS__ ); return -1; }while(0)
  if( NL < 1 ) FAIL( "model has no links" );
  if( NL > RKFD_MAX_LINK ) FAIL( "nlink %d exceeds the per-wave limit %d", NL, RKFD_MAX_LINK );
  if( ND > RKFD_MAX_DOF ) FAIL( "ndof %d exceeds the per-wave limit %d", ND, RKFD_MAX_DOF );
  if( NC > RKFD_MAX_CAND ) FAIL( "ncand %d exceeds the per-wave limit %d", NC, RKFD_MAX_CAND );
  if( max_rigid < 0 ) max_rigid = 0;
  if( 3*max_rigid > RKFD_MAX_ROWS ) FAIL( "3*max_rigid %d exceeds the per-wave limit %d", 3*max_rigid, RKFD_MAX_ROWS );
  for( int i=0; i<NL; i++ )
    if( m->parent[i] >= i ) FAIL( "link %d: parent index must be smaller than the link index", i );

  /* depth, levels */
  std::vector<int> depth( NL ), is_static( NL );
  int nlevel = 0;
  for( int i=0; i<NL; i++ ){
    const int p = m->parent[i];
    depth[i] = p < 0 ? 0 : depth[p]+1;
    is_static[i] = ( m->jtype[i] == RKFD_JOINT_FIXED ) && ( p < 0 || is_static[p] );
    if( depth[i]+1 > nlevel ) nlevel = depth[i]+1;
  }
  int nround = 0;
  while( ( 1 << nround ) < nlevel ) nround++;
  std::vector<int> level_off( nlevel+1, 0 ), level_link( NL );
  for( int i=0; i<NL; i++ ) level_off[depth[i]+1]++;
  for( int d=0; d<nlevel; d++ ) level_off[d+1] += level_off[d];
  {
    std::vector<int> cur( level_off.begin(), level_off.end()-1 );
    for( int i=0; i<NL; i++ ) level_link[cur[depth[i]]++] = i;
  }
  /* ancestor tables for pointer jumping */
  std::vector<int> anc( (size_t)( nround ? nround : 1 )*NL, -1 );
  for( int i=0; i<NL; i++ ) if( nround ) anc[i] = m->parent[i];
  for( int r=1; r<nround; r++ )
    for( int i=0; i<NL; i++ ){
      const int a = anc[(size_t)(r-1)*NL+i];
      anc[(size_t)r*NL+i] = a < 0 ? -1 : anc[(size_t)(r-1)*NL+a];
    }
  /* children CSR */
  std::vector<int> child_off( NL+1, 0 ), child_idx( NL );
  for( int i=0; i<NL; i++ ) if( m->parent[i] >= 0 ) child_off[m->parent[i]+1]++;
  for( int i=0; i<NL; i++ ) child_off[i+1] += child_off[i];
  {
    std::vector<int> cur( child_off.begin(), child_off.end()-1 );
    for( int i=0; i<NL; i++ ) if( m->parent[i] >= 0 ) child_idx[cur[m->parent[i]]++] = i;
  }
  /* ancestor at depth d */
  std::vector<int> pathlink( (size_t)NL*nlevel, -1 );
  for( int i=0; i<NL; i++ ){
    int a = i;
    while( a >= 0 ){ pathlink[(size_t)i*nlevel+depth[a]] = a; a = m->parent[a]; }
  }
  /* candidates */
  std::vector<int> cA( NC ), cB( NC ), cfo( NC ), cnf( NC ), cci( NC );
  std::vector<double> cv( (size_t)3*NC );
  for( int j=0; j<NC; j++ ){
    const int pr = m->cand_pair[j], sd = m->cand_side[j];
    const int shA = m->pair_shape[2*pr+sd], shB = m->pair_shape[2*pr+1-sd];
    cA[j] = m->shape_link[shA]; cB[j] = m->shape_link[shB];
    cfo[j] = m->shape_foff[shB]; cnf[j] = m->shape_foff[shB+1] - m->shape_foff[shB];
    cci[j] = m->pair_ci[pr];
    for( int k=0; k<3; k++ ) cv[3*j+k] = m->verts[3*m->cand_vert[j]+k];
  }
  const int nplane = m->nshape > 0 ? m->shape_foff[m->nshape] : 0;
  if( m->nci > 255 ) FAIL( "too many contact infos" );
  /* packed link / candidate info */
  std::vector<int> linfo( NL ), cinfo( NC );
  for( int i=0; i<NL; i++ )
    linfo[i] = RKFD_LI_PACK( m->parent[i], m->jtype[i], depth[i], is_static[i], m->mtype[i], m->dofoff[i] );
  for( int j=0; j<NC; j++ ){
    if( cnf[j] > 255 ) FAIL( "a collision shape has more than 255 faces" );
    cinfo[j] = cA[j] | ( cB[j] << 8 ) | ( cci[j] << 16 ) | ( cnf[j] << 24 );
  }
  /* sweep schedule: one iteration = up to 8 links of one level.  Lane-group slots are kept stable
   * along chains (a link takes the slot of its first child when possible) so that the sweeps can
   * hand data from one iteration to the next in registers:
   *   flag bit 0: the link's only child was processed in the NEXT iteration (sweep 2 runs the
   *               schedule backwards, so "previous" there) by the same slot
   *   flag bit 1: the link's parent was processed in the previous iteration by the same slot */
  std::vector<int> sched, pslot( NL, -1 ), fslot( NL, -1 );
  int nsched = 0, npool = 0, nfloat = 0;
  {
    std::vector<int> slot( NL, -1 ), iter( NL, -1 );
    std::vector<std::vector<int> > iters;          /* iteration -> 8 slots -> link */
    std::vector<int> level_first_iter( nlevel+1, 0 );
    /* assign slots bottom-up */
    std::vector<std::vector<std::vector<int> > > per_level( nlevel );
    for( int d=nlevel-1; d>=0; d-- ){
      const int n = level_off[d+1] - level_off[d];
      const int nch = ( n + 7 ) / 8;
      per_level[d].assign( nch, std::vector<int>( 8, -1 ) );
      std::vector<int> rest;
      for( int k=level_off[d]; k<level_off[d+1]; k++ ){
        const int i = level_link[k];
        int pref = -1;
        if( child_off[i+1] > child_off[i] ) pref = slot[child_idx[child_off[i]]];
        if( nch == 1 && pref >= 0 && per_level[d][0][pref] < 0 ){ per_level[d][0][pref] = i; slot[i] = pref; }
        else rest.push_back( i );
      }
      int c = 0, sidx = 0;
      for( size_t k=0; k<rest.size(); k++ ){
        while( per_level[d][c][sidx] >= 0 ){ sidx++; if( sidx == 8 ){ sidx = 0; c++; } }
        per_level[d][c][sidx] = rest[k]; slot[rest[k]] = sidx;
      }
    }
    for( int d=0; d<nlevel; d++ )
      for( size_t c=0; c<per_level[d].size(); c++ ){
        for( int g=0; g<8; g++ ) if( per_level[d][c][g] >= 0 ) iter[per_level[d][c][g]] = nsched;
        iters.push_back( per_level[d][c] );
        nsched++;
      }
    /* two empty iterations before and after the real ones: the sweeps prefetch records two
     * iterations ahead and read the padding instead of branching */
    /* pool slots: a non-float link with a parent needs its Ia staged in LDS unless the parent takes
     * it over in registers (flag bit 0 of the parent's record) */
    {
      std::vector<int> carried( NL, 0 );
      for( int i=0; i<NL; i++ ){
        const int nchild = child_off[i+1] - child_off[i];
        if( nchild == 1 ){
          const int ch = child_idx[child_off[i]];
          if( iter[ch] == iter[i]+1 && slot[ch] == slot[i] && m->jtype[ch] != RKFD_JOINT_FLOAT ) carried[ch] = 1;
        }
      }
      for( int i=0; i<NL; i++ ){
        pslot[i] = -1;
        if( m->parent[i] >= 0 && m->jtype[i] != RKFD_JOINT_FLOAT && !carried[i] ) pslot[i] = npool++;
        fslot[i] = -1;
        if( m->jtype[i] == RKFD_JOINT_FLOAT ) fslot[i] = nfloat++;
      }
    }
    for( int t=-2; t<nsched+2; t++ )
      for( int g=0; g<8; g++ ){
        int rec[4] = { -1, 0, 0, 0 };
        const int i = ( t >= 0 && t < nsched ) ? iters[t][g] : -1;
        if( i >= 0 ){
          const int nchild = child_off[i+1] - child_off[i];
          int flags = 0;
          if( nchild == 1 ){
            const int ch = child_idx[child_off[i]];
            if( iter[ch] == t+1 && slot[ch] == g && m->jtype[ch] != RKFD_JOINT_FLOAT ) flags |= 1;
          }
          if( m->parent[i] >= 0 && iter[m->parent[i]] == t-1 && slot[m->parent[i]] == g ) flags |= 2;
          if( nchild > 255 || pslot[i]+1 > 255 || fslot[i]+1 > 255 ) FAIL( "schedule record overflow" );
          rec[0] = i; rec[1] = linfo[i];
          rec[2] = nchild | ( flags << 8 ) | ( ( pslot[i]+1 ) << 16 ) | ( ( fslot[i]+1 ) << 24 );
          rec[3] = child_off[i];
        }
        sched.insert( sched.end(), rec, rec+4 );
      }
  }

  Blob b;
  rkfdDevModel dm;
  memset( &dm, 0, sizeof(dm) );
  dm.nlink = NL; dm.ndof = ND; dm.ncand = NC; dm.nlevel = nlevel; dm.nround = nround; dm.nci = m->nci;
  dm.solver = m->solver; dm.max_iter = m->max_iter; dm.maxrg = max_rigid;
  dm.dt = m->dt; dm.fric_w = m->friction_weight;
  dm.nsched = nsched; dm.npool = npool; dm.nfloat = nfloat;
  /* contact capacities */
  int maxact = NC < 16 ? NC : 16;
  if( max_rigid > maxact ) maxact = max_rigid < NC ? max_rigid : NC;
  int nside = 1;
  for( int j=0; j<NC; j++ )
    if( m->ci_type[cci[j]] == RKFD_CONTACT_RIGID && !is_static[cA[j]] && !is_static[cB[j]] ) nside = 2;
  dm.maxact = maxact; dm.nside = nside;
  if( nround > RKFD_MAX_ROUND ) FAIL( "tree too deep" );
  /* record offsets first (the vector may reallocate), then resolve */
  struct Ent { const void **slot; size_t off; };
  std::vector<Ent> ents;
#define PUT(field, src, bytes) ents.push_back( Ent{ (const void **)&dm.field, b.put( src, bytes ) } )
  PUT( parent, m->parent, sizeof(int)*NL ); PUT( jtype, m->jtype, sizeof(int)*NL );
  PUT( dofoff, m->dofoff, sizeof(int)*NL ); PUT( mtype, m->mtype, sizeof(int)*NL );
  PUT( depth, depth.data(), sizeof(int)*NL ); PUT( is_static, is_static.data(), sizeof(int)*NL );
  PUT( org, m->org, sizeof(double)*12*NL ); PUT( mass, m->mass, sizeof(double)*NL );
  PUT( com, m->com, sizeof(double)*3*NL ); PUT( inertia, m->inertia, sizeof(double)*9*NL );
  PUT( stiff, m->stiff, sizeof(double)*NL ); PUT( visc, m->visc, sizeof(double)*NL );
  PUT( coulomb, m->coulomb, sizeof(double)*NL ); PUT( sfric, m->sfric, sizeof(double)*NL );
  PUT( mot_k, m->mot_k, sizeof(double)*NL ); PUT( mot_admit, m->mot_admit, sizeof(double)*NL );
  PUT( mot_vmax, m->mot_vmax, sizeof(double)*NL ); PUT( mot_vmin, m->mot_vmin, sizeof(double)*NL );
  PUT( mot_gear, m->mot_gear, sizeof(double)*NL ); PUT( mot_inertia, m->mot_inertia, sizeof(double)*NL );
  PUT( anc, anc.data(), sizeof(int)*anc.size() );
  PUT( level_off, level_off.data(), sizeof(int)*( nlevel+1 ) ); PUT( level_link, level_link.data(), sizeof(int)*NL );
  PUT( child_off, child_off.data(), sizeof(int)*( NL+1 ) ); PUT( child_idx, child_idx.data(), sizeof(int)*NL );
  PUT( pathlink, pathlink.data(), sizeof(int)*pathlink.size() );
  PUT( linfo, linfo.data(), sizeof(int)*NL ); PUT( cinfo, cinfo.data(), sizeof(int)*NC );
  PUT( sched, sched.data(), sizeof(int)*sched.size() );
  PUT( pslot, pslot.data(), sizeof(int)*NL );
  PUT( cand_linkA, cA.data(), sizeof(int)*NC ); PUT( cand_linkB, cB.data(), sizeof(int)*NC );
  PUT( cand_foff, cfo.data(), sizeof(int)*NC ); PUT( cand_nf, cnf.data(), sizeof(int)*NC );
  PUT( cand_ci, cci.data(), sizeof(int)*NC ); PUT( cand_vert, cv.data(), sizeof(double)*3*NC );
  PUT( planes, m->planes, sizeof(double)*4*nplane );
  PUT( ci_type, m->ci_type, sizeof(int)*m->nci );
  PUT( ci_sf, m->ci_sf, sizeof(double)*m->nci ); PUT( ci_kf, m->ci_kf, sizeof(double)*m->nci );
  PUT( ci_k, m->ci_k, sizeof(double)*m->nci ); PUT( ci_l, m->ci_l, sizeof(double)*m->nci );
  PUT( ci_e, m->ci_e, sizeof(double)*m->nci ); PUT( ci_v, m->ci_v, sizeof(double)*m->nci );
#undef PUT
  out->bytes = b.buf.size();
  out->blob = malloc( out->bytes );
  if( !out->blob ) FAIL( "out of memory" );
  memcpy( out->blob, b.buf.data(), out->bytes );
  for( size_t k=0; k<ents.size(); k++ ) *ents[k].slot = (const char *)out->blob + ents[k].off;
  out->dm = dm;
  /* LDS bytes one instance needs (must match rkfd_lds_carve in rkfd_device.h) */
  {
    const size_t M = 3*(size_t)max_rigid;
    size_t stage = (size_t)14*NL + (size_t)36*npool;            /* inertia staging + Ia pool ...   */
    if( M*(M+1) > stage ) stage = M*(M+1);                       /* ... shared with the MLCP matrix */
    const size_t dbl = (size_t)4*ND + (size_t)NL*( 8*6 + 4 ) + stage + (size_t)48*nfloat
                     + (size_t)maxact*18 + (size_t)NC*6 + 2*M + (size_t)nside*nlevel*M + 2*(size_t)NL;
    const size_t ints = (size_t)5*NC + 8 + (size_t)ND + (size_t)NL      /* act typ lrg lel tgt, cnt, dofkind, pivt */
                      + 3*(size_t)NL + 3*(size_t)NC + ( max_rigid > 0 ? ( (size_t)NL*nlevel + 3 )/4 : 0 ); /* LI, CHI, PSL, CIp, CFO, asl, PL (bytes) */
    out->lds_bytes = dbl*sizeof(double) + ints*sizeof(int);
    out->lds_bytes = ( out->lds_bytes + 15 ) & ~(size_t)15;
  }
  return 0;
#undef FAIL
}

extern "C" void rkfd_devmodel_free(rkfdDevModelHost *h)
{
  if( h ){ free( h->blob ); h->blob = NULL; }
}

extern "C" void rkfd_devmodel_rebase(rkfdDevModel *dm, const void *from, const void *to)
{
  const ptrdiff_t d = (const char *)to - (const char *)from;
#define RB(f) dm->f = (decltype(dm->f))( (const char *)dm->f + d )
  RB(parent); RB(jtype); RB(dofoff); RB(mtype); RB(depth); RB(is_static);
  RB(org); RB(mass); RB(com); RB(inertia); RB(stiff); RB(visc); RB(coulomb); RB(sfric);
  RB(mot_k); RB(mot_admit); RB(mot_vmax); RB(mot_vmin); RB(mot_gear); RB(mot_inertia);
  RB(anc); RB(level_off); RB(level_link); RB(child_off); RB(child_idx); RB(pathlink); RB(linfo); RB(sched); RB(cinfo); RB(pslot);
  RB(cand_linkA); RB(cand_linkB); RB(cand_foff); RB(cand_nf); RB(cand_ci); RB(cand_vert); RB(planes);
  RB(ci_type); RB(ci_sf); RB(ci_kf); RB(ci_k); RB(ci_l); RB(ci_e); RB(ci_v);
#undef RB
}
