/* rkfd_devmodel.cpp - host-side construction of the device model tables from an rkfdModel.
 * Pure host C++ (no HIP): used by the C-ABI (rkfd_capi.hip), which copies the blob to HBM,
 * and by the lane emulator under tests/emu.
 */
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#include <utility>
#include "rkfd_model.h"
#include "rkfd_devmodel.h"
#include "rkfd_devmodel_host.h"

/* test switch, process-wide, applied to the batches created after the call: XORed into the variant mask (see below) */
static int g_debug_variants = 0;
extern "C" int rkfdDebugVariants(int mask){ const int old = g_debug_variants; g_debug_variants = mask & ( 4 | 8 | 32 ); return old; }

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
  return rkfd_devmodel_build_w( m, max_rigid, 8, out, err, errlen );
}
extern "C" int rkfd_devmodel_build_w(const rkfdModel *m, int max_rigid, const int NG, rkfdDevModelHost *out, char *err, int errlen)
{
  const int NLm = m->nlink, ND = m->ndof, NC = m->ncand;
  std::vector<int> R_parent, R_jtype, R_dofoff, R_mtype;
  std::vector<double> R_org;
#define FAIL(...) do{ if( err ) snprintf( err, errlen, __VA_ARGS__ ); return -1; }while(0)
  if( NLm < 1 ) FAIL( "model has no links" );
  if( ND > RKFD_MAX_DOF ) FAIL( "ndof %d exceeds the per-wave limit %d", ND, RKFD_MAX_DOF );
  if( NC > RKFD_MAX_CAND ) FAIL( "ncand %d exceeds the per-wave limit %d", NC, RKFD_MAX_CAND );
  if( max_rigid < 0 ) max_rigid = 0;
  if( 3*max_rigid > RKFD_MAX_ROWS ) FAIL( "3*max_rigid %d exceeds the per-wave limit %d", 3*max_rigid, RKFD_MAX_ROWS );
  int has_brf = 0;
  for( int i=0; i<NLm; i++ ){
    if( m->parent[i] >= i ) FAIL( "link %d: parent index must be smaller than the link index", i );
    if( m->jtype[i] == RKFD_JOINT_BRFLOAT ) has_brf = 1;
  }
  if( has_brf ){
    /* breakable float joints (device/rkfd_dev_brf.h): whatever hangs on one must be attached by breakable float or fixed joints
     * only - then an unbroken joint carries a RIGID subtree, and the wrench it transmits is that body's momentum balance */
    for( int i=0; i<NLm; i++ ){
      if( m->jtype[i] == RKFD_JOINT_BRFLOAT || m->jtype[i] == RKFD_JOINT_FIXED ) continue;
      for( int a=m->parent[i]; a>=0; a=m->parent[a] )
        if( m->jtype[a] == RKFD_JOINT_BRFLOAT )
          FAIL( "link %d hangs below the breakable float joint of link %d on a joint that can move: only breakable float and fixed joints may follow a breakable float joint on the device", i, a );
    }
  }

  /* ---- merge rigidly attached links ------------------------------------------------------
   * A link on a FIXED joint moves with its parent: it is folded into the nearest ancestor that is
   * not such a link (composite rigid body: summed mass, combined centre of mass and inertia; its
   * children, shapes and candidate vertices are re-expressed in that ancestor's frame).  The
   * dynamics are unchanged, the sweeps get fewer links and levels.  Fixed ROOT links stay. */
  std::vector<int> rep( NLm ), orig;                 /* model link -> device link, and back      */
  std::vector<double> Trep( (size_t)12*NLm );        /* frame of model link i in its device link  */
  struct Acc { double m; double mc[3]; std::vector<int> parts; };
  std::vector<Acc> accs;
  R_parent.clear(); R_jtype.clear(); R_dofoff.clear(); R_mtype.clear(); R_org.clear();
  for( int i=0; i<NLm; i++ ){
    const double *o = &m->org[12*i];
    const int p = m->parent[i];
    if( m->jtype[i] == RKFD_JOINT_FIXED && p >= 0 ){
      /* T(rep<-i) = T(rep<-p) o org_i */
      const double *Tp = &Trep[12*p];
      double *T = &Trep[12*i];
      for( int a=0; a<3; a++ ){
        for( int b=0; b<3; b++ ) T[3*a+b] = Tp[3*a]*o[b] + Tp[3*a+1]*o[3+b] + Tp[3*a+2]*o[6+b];
        T[9+a] = Tp[9+a] + Tp[3*a]*o[9] + Tp[3*a+1]*o[10] + Tp[3*a+2]*o[11];
      }
      rep[i] = rep[p];
      accs[rep[i]].parts.push_back( i );
    } else {
      const bool sph = m->jtype[i] == RKFD_JOINT_SPHER;
      if( sph ){
        /* two massless pseudo-links in front of the real one (see RKFD_DJT_SPH*): the first carries the org frame */
        for( int q=0; q<2; q++ ){
          const int rq = (int)orig.size();
          orig.push_back( i );
          Acc a; a.m = 0; a.mc[0] = a.mc[1] = a.mc[2] = 0;
          accs.push_back( a );
          R_parent.push_back( q == 0 ? ( p < 0 ? -1 : rep[p] ) : rq-1 );
          R_jtype.push_back( q == 0 ? RKFD_DJT_SPHX : RKFD_DJT_SPHY ); R_dofoff.push_back( m->dofoff[i]+q ); R_mtype.push_back( RKFD_MOTOR_NONE );
          double oo[12];
          if( q == 1 || p < 0 ) for( int k=0; k<12; k++ ) oo[k] = q == 1 ? ( ( k == 0 || k == 4 || k == 8 ) ? 1.0 : 0.0 ) : o[k];
          else {
            const double *Tp = &Trep[12*p];
            for( int a=0; a<3; a++ ){
              for( int b=0; b<3; b++ ) oo[3*a+b] = Tp[3*a]*o[b] + Tp[3*a+1]*o[3+b] + Tp[3*a+2]*o[6+b];
              oo[9+a] = Tp[9+a] + Tp[3*a]*o[9] + Tp[3*a+1]*o[10] + Tp[3*a+2]*o[11];
            }
          }
          R_org.insert( R_org.end(), oo, oo+12 );
        }
      }
      const int r = (int)orig.size();
      rep[i] = r; orig.push_back( i );
      double *T = &Trep[12*i];
      for( int k=0; k<12; k++ ) T[k] = ( k == 0 || k == 4 || k == 8 ) ? 1.0 : 0.0;
      Acc a; a.m = 0; a.mc[0] = a.mc[1] = a.mc[2] = 0; a.parts.push_back( i );
      accs.push_back( a );
      R_parent.push_back( sph ? r-1 : ( p < 0 ? -1 : rep[p] ) );
      R_jtype.push_back( sph ? RKFD_DJT_SPHZ : ( m->jtype[i] == RKFD_JOINT_BRFLOAT ? (int)RKFD_JOINT_FLOAT : m->jtype[i] ) ); R_dofoff.push_back( m->dofoff[i] + ( sph ? 2 : 0 ) ); R_mtype.push_back( sph ? RKFD_MOTOR_NONE : m->mtype[i] );
      if( sph ){
        const double id[12] = { 1,0,0, 0,1,0, 0,0,1, 0,0,0 };
        R_org.insert( R_org.end(), id, id+12 );
        continue;
      }
      /* org' = T(rep[p]<-p) o org_i */
      double oo[12];
      if( p < 0 ) for( int k=0; k<12; k++ ) oo[k] = o[k];
      else {
        const double *Tp = &Trep[12*p];
        for( int a=0; a<3; a++ ){
          for( int b=0; b<3; b++ ) oo[3*a+b] = Tp[3*a]*o[b] + Tp[3*a+1]*o[3+b] + Tp[3*a+2]*o[6+b];
          oo[9+a] = Tp[9+a] + Tp[3*a]*o[9] + Tp[3*a+1]*o[10] + Tp[3*a+2]*o[11];
        }
      }
      R_org.insert( R_org.end(), oo, oo+12 );
    }
  }
  const int NL = (int)orig.size();
  if( NG != 8 && NG != 4 ) FAIL( "ngroup must be 8 or 4" );
  if( NL > RKFD_MAX_LINK ) FAIL( "nlink %d (after merging fixed links) exceeds the per-wave limit %d", NL, RKFD_MAX_LINK );
  std::vector<double> R_mass( NL, 0.0 ), R_com( (size_t)3*NL, 0.0 ), R_inertia( (size_t)9*NL, 0.0 );
  for( int r=0; r<NL; r++ ){
    double M = 0, mc[3] = {0,0,0};
    std::vector<double> cs;                         /* part centres of mass in the device link frame */
    for( size_t q=0; q<accs[r].parts.size(); q++ ){
      const int i = accs[r].parts[q];
      const double *T = &Trep[12*i], *c = &m->com[3*i];
      double cc[3];
      for( int a=0; a<3; a++ ) cc[a] = T[9+a] + T[3*a]*c[0] + T[3*a+1]*c[1] + T[3*a+2]*c[2];
      cs.insert( cs.end(), cc, cc+3 );
      M += m->mass[i];
      for( int a=0; a<3; a++ ) mc[a] += m->mass[i]*cc[a];
    }
    double com[3] = {0,0,0};
    if( M > 0 ) for( int a=0; a<3; a++ ) com[a] = mc[a]/M;
    else if( !cs.empty() ) for( int a=0; a<3; a++ ) com[a] = cs[a];
    double I[9] = {0,0,0,0,0,0,0,0,0};
    for( size_t q=0; q<accs[r].parts.size(); q++ ){
      const int i = accs[r].parts[q];
      const double *T = &Trep[12*i], *Ii = &m->inertia[9*i];
      double RI[9], RIRt[9];
      for( int a=0; a<3; a++ ) for( int b=0; b<3; b++ ) RI[3*a+b] = T[3*a]*Ii[b] + T[3*a+1]*Ii[3+b] + T[3*a+2]*Ii[6+b];
      for( int a=0; a<3; a++ ) for( int b=0; b<3; b++ ) RIRt[3*a+b] = RI[3*a]*T[3*b] + RI[3*a+1]*T[3*b+1] + RI[3*a+2]*T[3*b+2];
      const double d[3] = { cs[3*q]-com[0], cs[3*q+1]-com[1], cs[3*q+2]-com[2] };
      const double d2 = d[0]*d[0] + d[1]*d[1] + d[2]*d[2];
      for( int a=0; a<3; a++ ) for( int b=0; b<3; b++ )
        I[3*a+b] += RIRt[3*a+b] + m->mass[i]*( ( a == b ? d2 : 0.0 ) - d[a]*d[b] );
    }
    R_mass[r] = M;
    for( int a=0; a<3; a++ ) R_com[3*r+a] = com[a];
    for( int a=0; a<9; a++ ) R_inertia[9*r+a] = I[a];
  }
  /* 1-DoF joint parameters of the surviving links */
  std::vector<double> R_stiff( NL ), R_visc( NL ), R_coulomb( NL ), R_sfric( NL ), R_mot_k( NL ), R_mot_admit( NL ),
                      R_mot_vmax( NL ), R_mot_vmin( NL ), R_mot_gear( NL ), R_mot_inertia( NL );
  for( int r=0; r<NL; r++ ){
    const int i = orig[r];
    R_stiff[r] = m->stiff[i]; R_visc[r] = m->visc[i]; R_coulomb[r] = m->coulomb[i]; R_sfric[r] = m->sfric[i];
    R_mot_k[r] = m->mot_k[i]; R_mot_admit[r] = m->mot_admit[i]; R_mot_vmax[r] = m->mot_vmax[i]; R_mot_vmin[r] = m->mot_vmin[i];
    R_mot_gear[r] = m->mot_gear[i];
    /* the device keeps the inertia reflected through the gear (same product order as the oracle's), 0 without a DC motor */
    R_mot_inertia[r] = ( ( m->jtype[i] == RKFD_JOINT_REVOL || m->jtype[i] == RKFD_JOINT_PRISM ) && m->mtype[i] == RKFD_MOTOR_DC ) ? m->mot_inertia[i]*m->mot_gear[i]*m->mot_gear[i] : 0.0;
  }

  /* breakable float joints of the surviving links */
  std::vector<int> R_brf( NL, 0 );
  std::vector<double> R_brk_f( NL, 0.0 ), R_brk_t( NL, 0.0 );
  for( int r=0; r<NL; r++ ){
    const int i = orig[r];
    if( m->jtype[i] == RKFD_JOINT_BRFLOAT && R_jtype[r] == RKFD_JOINT_FLOAT ){ R_brf[r] = 1; R_brk_f[r] = m->brk_f[i]; R_brk_t[r] = m->brk_t[i]; }
  }

  /* depth, levels */
  std::vector<int> depth( NL ), is_static( NL );
  int nlevel = 0;
  for( int i=0; i<NL; i++ ){
    const int p = R_parent[i];
    depth[i] = p < 0 ? 0 : depth[p]+1;
    is_static[i] = ( R_jtype[i] == RKFD_JOINT_FIXED ) && ( p < 0 || is_static[p] );
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
  for( int i=0; i<NL; i++ ) if( nround ) anc[i] = R_parent[i];
  for( int r=1; r<nround; r++ )
    for( int i=0; i<NL; i++ ){
      const int a = anc[(size_t)(r-1)*NL+i];
      anc[(size_t)r*NL+i] = a < 0 ? -1 : anc[(size_t)(r-1)*NL+a];
    }
  /* children CSR */
  std::vector<int> child_off( NL+1, 0 ), child_idx( NL );
  for( int i=0; i<NL; i++ ) if( R_parent[i] >= 0 ) child_off[R_parent[i]+1]++;
  for( int i=0; i<NL; i++ ) child_off[i+1] += child_off[i];
  {
    std::vector<int> cur( child_off.begin(), child_off.end()-1 );
    for( int i=0; i<NL; i++ ) if( R_parent[i] >= 0 ) child_idx[cur[R_parent[i]]++] = i;
  }
  /* ancestor at depth d */
  std::vector<int> pathlink( (size_t)NL*( nlevel+3 ), -1 );
  for( int i=0; i<NL; i++ ){
    int a = i;
    while( a >= 0 ){ pathlink[(size_t)i*nlevel+depth[a]] = a; a = R_parent[a]; }
  }
  /* row nlevel: the link where a force applied to link i stops propagating upwards - the nearest
   * float joint at or above i, else the root (-1 for a link that cannot move) */
  for( int i=0; i<NL; i++ ){
    int a = i;
    while( R_jtype[a] != RKFD_JOINT_FLOAT && R_parent[a] >= 0 ) a = R_parent[a];
    pathlink[(size_t)NL*nlevel+i] = is_static[i] ? -1 : a;
  }
  /* candidates: owner / other link after merging, vertex in the owner device link's frame;
   * planes of every shape re-expressed in its device link's frame */
  const int nplane_m = m->nshape > 0 ? m->shape_foff[m->nshape] : 0;
  std::vector<double> planes_d( (size_t)4*( nplane_m ? nplane_m : 1 ) );
  for( int sh=0; sh<m->nshape; sh++ ){
    const double *T = &Trep[12*m->shape_link[sh]];
    for( int f=m->shape_foff[sh]; f<m->shape_foff[sh+1]; f++ ){
      const double *pl = &m->planes[4*f];
      double n[3];
      for( int a=0; a<3; a++ ) n[a] = T[3*a]*pl[0] + T[3*a+1]*pl[1] + T[3*a+2]*pl[2];
      planes_d[4*f] = n[0]; planes_d[4*f+1] = n[1]; planes_d[4*f+2] = n[2];
      planes_d[4*f+3] = pl[3] + n[0]*T[9] + n[1]*T[10] + n[2]*T[11];
    }
  }
  std::vector<int> cA( NC ), cB( NC ), cfo( NC ), cnf( NC ), cci( NC );
  std::vector<double> cv( (size_t)3*NC );
  for( int j=0; j<NC; j++ ){
    const int pr = m->cand_pair[j], sd = m->cand_side[j];
    const int shA = m->pair_shape[2*pr+sd], shB = m->pair_shape[2*pr+1-sd];
    cA[j] = rep[m->shape_link[shA]]; cB[j] = rep[m->shape_link[shB]];
    cfo[j] = m->shape_foff[shB]; cnf[j] = m->shape_foff[shB+1] - m->shape_foff[shB];
    if( cfo[j] > 65535 ) FAIL( "more than 65535 shape faces (the device keeps a candidate's first plane in 16 bits)" );
    cci[j] = m->pair_ci[pr];
    const double *T = &Trep[12*m->shape_link[shA]], *v = &m->verts[3*m->cand_vert[j]];
    for( int a=0; a<3; a++ ) cv[3*j+a] = T[9+a] + T[3*a]*v[0] + T[3*a+1]*v[1] + T[3*a+2]*v[2];
  }
  /* broad phase: a sphere around the other shape (its vertices, in its device link's frame) that a candidate vertex
   * must enter before it is tested against the shape's planes.  The margin (0.1 % + 1 mm) is far above the contact
   * tolerance, so the cull never changes a result */
  std::vector<double> cbs( (size_t)4*( NC ? NC : 1 ), 0.0 );
  for( int j=0; j<NC; j++ ){
    const int pr = m->cand_pair[j], sd = m->cand_side[j];
    const int shB = m->pair_shape[2*pr+1-sd];
    const double *T = &Trep[12*m->shape_link[shB]];
    double lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
    std::vector<double> w;
    for( int v=m->shape_voff[shB]; v<m->shape_voff[shB+1]; v++ ){
      const double *p = &m->verts[3*v];
      for( int a=0; a<3; a++ ){
        const double x = T[9+a] + T[3*a]*p[0] + T[3*a+1]*p[1] + T[3*a+2]*p[2];
        w.push_back( x ); if( x < lo[a] ) lo[a] = x; if( x > hi[a] ) hi[a] = x;
      }
    }
    double r2 = 0;
    for( int a=0; a<3; a++ ) cbs[4*j+a] = w.empty() ? 0.0 : 0.5*( lo[a]+hi[a] );
    for( size_t v=0; v+2<w.size(); v+=3 ){
      double d2 = 0;
      for( int a=0; a<3; a++ ){ const double d = w[v+a]-cbs[4*j+a]; d2 += d*d; }
      if( d2 > r2 ) r2 = d2;
    }
    const double r = sqrt( r2 )*1.001 + 1e-3;
    cbs[4*j+3] = w.empty() ? 1e300 : r*r;
  }
  std::vector<double> refT( (size_t)12*( NC ? NC : 1 ), 0.0 );
  for( int j=0; j<NC; j++ ){
    const int pr = m->cand_pair[j], sd = m->cand_side[j];
    memcpy( &refT[12*j], &Trep[12*m->shape_link[m->pair_shape[2*pr+1-sd]]], sizeof(double)*12 );
  }
  /* slide-mode cells of the candidates */
  int has_slide = 0;
  for( int sh=0; sh<m->nshape; sh++ ) if( m->shape_slide_mode && m->shape_slide_mode[sh] ) has_slide = 1;
  std::vector<int> csm( has_slide ? (size_t)2*NC : 1, 0 );
  std::vector<double> csp( has_slide ? (size_t)14*NC : 1, 0.0 );
  if( has_slide ){
    for( int j=0; j<NC; j++ ){
      const int pr = m->cand_pair[j], sd = m->cand_side[j];
      for( int k=0; k<2; k++ ){            /* k = 0: the cell that owns the vertex, 1: the other cell */
        const int i = k == 0 ? sd : 1-sd;  /* index of that cell in the pair */
        const int sh = m->pair_shape[2*pr+i];
        const int isown = k == 0;
        /* the anchor drift is rotated into the frame of cell[ isown ? 1 : 0 ] (reference src/rkfd_util.c:232) */
        const int tgt_is_own = ( isown ? 1 : 0 ) == sd;
        csm[2*j+k] = m->shape_slide_mode[sh] ? ( tgt_is_own ? 3 : 1 ) : 0;
        const double *T = &Trep[12*m->shape_link[sh]], *ax = &m->shape_slide_axis[3*sh];
        double *o = &csp[14*j+7*k];
        o[0] = m->shape_slide_vel[sh];
        for( int a=0; a<3; a++ ){ o[1+a] = T[3*a]*ax[0] + T[3*a+1]*ax[1] + T[3*a+2]*ax[2]; o[4+a] = T[9+a]; }
      }
    }
  }
  const int nplane = m->nshape > 0 ? m->shape_foff[m->nshape] : 0;
  if( m->nci > 255 ) FAIL( "too many contact infos" );
  /* packed link / candidate info */
  std::vector<int> linfo( NL ), cinfo( NC );
  for( int i=0; i<NL; i++ )
    linfo[i] = RKFD_LI_PACK( R_parent[i], R_jtype[i], depth[i], is_static[i], R_mtype[i], R_dofoff[i] );
  for( int j=0; j<NC; j++ ){
    if( cnf[j] > 4095 ) FAIL( "a collision shape has more than 4095 face planes" );
    if( cci[j] > 63 ) FAIL( "more than 63 contact-info entries" );
    cinfo[j] = RKFD_CI_PACK( cA[j], cB[j], cci[j], cnf[j] );
  }
  /* Volume plugin: the rigid pairs with the face loops of their (convex) shapes - for every plane of a shape the
   * vertices lying on it, counter-clockwise seen from outside, in the device link's frame (the oracle builds the same
   * loops: oracle/rkfd_oracle_volume.h vol_prepare).  A coplanar duplicate of an earlier plane gets no loop. */
  std::vector<int> vol_pair, vol_loop;
  std::vector<double> vol_lplane, vol_lvert, vol_slide;
  int vol_npair = 0, vol_np = 0, vol_ncp = 0, vol_pv = 0, vol_nf = 0;
  if( m->solver == RKFD_SOLVER_VOLUME ){
    std::vector<int> sh_l0( m->nshape, -1 ), sh_nl( m->nshape, 0 ), sh_raw( m->nshape, 0 );
    int maxloop = 0, maxfaces = 0, nclip = 0;
    for( int pr=0; pr<m->npair; pr++ ){
      if( m->ci_type[m->pair_ci[pr]] != RKFD_CONTACT_RIGID ) continue;
      for( int sd=0; sd<2; sd++ ){
        const int sh = m->pair_shape[2*pr+sd];
        if( sh_l0[sh] >= 0 ) continue;
        const double *T = &Trep[12*m->shape_link[sh]];
        for( int f=m->shape_foff[sh]; f<m->shape_foff[sh+1] && !sh_raw[sh]; f++ )
          for( int v=m->shape_voff[sh]; v<m->shape_voff[sh+1]; v++ ){
            const double *pl = &m->planes[4*f], *x = &m->verts[3*v];
            if( pl[0]*x[0] + pl[1]*x[1] + pl[2]*x[2] - pl[3] > 1e-9 ){ sh_raw[sh] = 1; break; }
          }
        sh_l0[sh] = (int)vol_loop.size()/2;
        if( sh_raw[sh] ){
          /* a shape that is not convex (the reference's humanoid mighty.ztk: its body meshes): the intersection volume is formed
           * by clipping CONVEX shapes, so its pairs cannot be clipped - they are GUARDED instead: the plugin's own collision test
           * (a vertex of one shape behind every face plane of the other, "Vert") runs for them, and a hit is reported as status 4
           * instead of being solved.  For that test: every plane of the shape, and its vertices as the "loop" of the last plane. */
          const int nf = m->shape_foff[sh+1] - m->shape_foff[sh], nv = m->shape_voff[sh+1] - m->shape_voff[sh];
          const int vbase = (int)vol_lvert.size()/3;
          for( int v=m->shape_voff[sh]; v<m->shape_voff[sh+1]; v++ ){
            const double *x = &m->verts[3*v];
            for( int a=0; a<3; a++ ) vol_lvert.push_back( T[9+a] + T[3*a]*x[0] + T[3*a+1]*x[1] + T[3*a+2]*x[2] );
          }
          for( int f=0; f<nf; f++ ){
            vol_loop.push_back( vbase ); vol_loop.push_back( f == nf-1 ? nv : 0 );
            for( int a=0; a<4; a++ ) vol_lplane.push_back( planes_d[4*( m->shape_foff[sh]+f )+a] );
          }
          sh_nl[sh] = nf;
          continue;
        }
        for( int f=m->shape_foff[sh]; f<m->shape_foff[sh+1]; f++ ){
          const double *pl = &m->planes[4*f];
          bool dup = false;
          for( int f2=m->shape_foff[sh]; f2<f; f2++ ){
            const double *p2 = &m->planes[4*f2];
            if( fabs( pl[0]-p2[0] ) < 1e-9 && fabs( pl[1]-p2[1] ) < 1e-9 && fabs( pl[2]-p2[2] ) < 1e-9 && fabs( pl[3]-p2[3] ) < 1e-9 ) dup = true;
          }
          if( dup ) continue;
          std::vector<int> idx;
          for( int v=m->shape_voff[sh]; v<m->shape_voff[sh+1]; v++ ){
            const double *x = &m->verts[3*v];
            if( !( fabs( pl[0]*x[0] + pl[1]*x[1] + pl[2]*x[2] - pl[3] ) < 1e-9 ) ) continue;
            bool same = false;
            for( size_t i=0; i<idx.size(); i++ ){
              const double *a = &m->verts[3*idx[i]];
              if( fabs( a[0]-x[0] ) < 1e-12 && fabs( a[1]-x[1] ) < 1e-12 && fabs( a[2]-x[2] ) < 1e-12 ) same = true;
            }
            if( !same ) idx.push_back( v );
          }
          if( idx.size() < 3 ) continue;
          /* order by angle about the centroid in the plane's own basis (tangent 1 = the unit vector of the smallest |component| made
           * orthogonal to n, tangent 2 = n x t1: the oracle's ortho_space) */
          double c[3] = {0,0,0}, e[3] = {0,0,0}, t1[3], t2[3];
          for( size_t i=0; i<idx.size(); i++ ) for( int a=0; a<3; a++ ) c[a] += ( 1.0/idx.size() )*m->verts[3*idx[i]+a];
          int k = 0;
          if( fabs( pl[1] ) < fabs( pl[k] ) ) k = 1;
          if( fabs( pl[2] ) < fabs( pl[k] ) ) k = 2;
          e[k] = 1.0;
          const double dd = e[0]*pl[0] + e[1]*pl[1] + e[2]*pl[2];
          for( int a=0; a<3; a++ ) t1[a] = e[a] - dd*pl[a];
          const double l = sqrt( t1[0]*t1[0] + t1[1]*t1[1] + t1[2]*t1[2] );
          for( int a=0; a<3; a++ ) t1[a] /= l;
          t2[0] = pl[1]*t1[2]-pl[2]*t1[1]; t2[1] = pl[2]*t1[0]-pl[0]*t1[2]; t2[2] = pl[0]*t1[1]-pl[1]*t1[0];
          std::vector<std::pair<double,int> > ang;
          for( size_t i=0; i<idx.size(); i++ ){
            const double *x = &m->verts[3*idx[i]];
            const double d[3] = { x[0]-c[0], x[1]-c[1], x[2]-c[2] };
            ang.push_back( std::make_pair( atan2( d[0]*t2[0]+d[1]*t2[1]+d[2]*t2[2], d[0]*t1[0]+d[1]*t1[1]+d[2]*t1[2] ), idx[i] ) );
          }
          std::stable_sort( ang.begin(), ang.end(), []( const std::pair<double,int> &a, const std::pair<double,int> &b ){ return a.first < b.first; } );
          vol_loop.push_back( (int)vol_lvert.size()/3 ); vol_loop.push_back( (int)ang.size() );
          for( size_t i=0; i<ang.size(); i++ ){
            const double *x = &m->verts[3*ang[i].second];
            for( int a=0; a<3; a++ ) vol_lvert.push_back( T[9+a] + T[3*a]*x[0] + T[3*a+1]*x[1] + T[3*a+2]*x[2] );
          }
          for( int a=0; a<4; a++ ) vol_lplane.push_back( planes_d[4*f+a] );
          if( (int)ang.size() > maxloop ) maxloop = (int)ang.size();
          sh_nl[sh]++;
        }
        if( sh_nl[sh] < 4 ) FAIL( "Volume plugin: shape %d of a rigid pair is not a closed polyhedron (%d faces found)", sh, sh_nl[sh] );
      }
      const int shA = m->pair_shape[2*pr], shB = m->pair_shape[2*pr+1];
      /* slide-mode cells (rkFDLinkAddSlideVel, reference src/rkfd_util.c:26-40; the plugin's friction fix-ups see them, its 6-D
       * velocity does not, :83-85): per side the speed, the axis and the origin of the shape's model link, in the device link's frame */
      int smode = 0;
      for( int sd=0; sd<2; sd++ ){
        const int sh = m->pair_shape[2*pr+sd];
        const double *T = &Trep[12*m->shape_link[sh]];
        const bool on = m->shape_slide_mode && m->shape_slide_mode[sh];
        if( on ) smode |= 1 << sd;
        double o[8] = { on ? m->shape_slide_vel[sh] : 0.0, 0, 0, 0, T[9], T[10], T[11], 0 };
        if( on ){ const double *ax = &m->shape_slide_axis[3*sh]; for( int a=0; a<3; a++ ) o[1+a] = T[3*a]*ax[0] + T[3*a+1]*ax[1] + T[3*a+2]*ax[2]; }
        vol_slide.insert( vol_slide.end(), o, o+8 );
      }
      /* (bit 2 of the mode word: a guarded pair - a shape that is not convex, or more faces together than lanes) */
      const bool guard = sh_raw[shA] || sh_raw[shB] || sh_nl[shA] + sh_nl[shB] > RKFD_WAVE;
      const int rec[8] = { rep[m->shape_link[shA]], rep[m->shape_link[shB]], m->pair_ci[pr], sh_l0[shA], sh_nl[shA], sh_l0[shB], sh_nl[shB], smode | ( guard ? 4 : 0 ) };
      vol_pair.insert( vol_pair.end(), rec, rec+8 );
      vol_npair++;
      if( guard ) continue;
      nclip++;
      if( sh_nl[shA] + sh_nl[shB] > maxfaces ) maxfaces = sh_nl[shA] + sh_nl[shB];
      const int big = sh_nl[shA] > sh_nl[shB] ? sh_nl[shA] : sh_nl[shB];
      if( maxloop + big > vol_pv ) vol_pv = maxloop + big;
    }
    if( vol_npair > 0 && nclip == 0 )
      FAIL( "Volume plugin: no rigid pair of this world can be clipped on the device (convex shapes with at most %d faces together)", RKFD_WAVE );
    if( vol_npair > 0 ){
      /* capacities: pairs in collision at once (the caller's max_rigid, at most 10: six unknowns each, one per lane),
       * contact-plane conditions per pair (every face of the two shapes can give one), constraints <= 64 */
      vol_np = max_rigid > 0 ? ( max_rigid < vol_npair ? max_rigid : vol_npair ) : 0;
      if( vol_np > 10 ) vol_np = 10;
      if( vol_np > 0 ){
        /* contact-plane conditions per pair: the contact polygon of two boxes (12 faces together) has at most eight edges;
         * rounder shapes (a cylinder on its end: one edge per side, one more when it tilts) get one per face of the pair - as far
         * as the simplex's 192 register-resident columns (pyramid x conditions + 7), the QP's 64 constraints and the LDS of the
         * solve's workspace (24 KB) allow: 23 for one pair of a 16-sided cylinder and a box.
         * More at run time is reported (status 2). */
        const int pyr = m->pyramid > 0 ? m->pyramid : 8;
        vol_ncp = maxfaces <= 12 ? ( maxfaces < 8 ? maxfaces : 8 ) : maxfaces;
        while( vol_ncp > 4 && ( vol_np*( 1+vol_ncp ) > RKFD_WAVE || pyr*vol_ncp + 7 > 192 ) ) vol_ncp--;
        while( vol_ncp > 8 && RKFD_VOL_LDS_SOL( vol_np, vol_ncp, pyr ) > 3072 ) vol_ncp--;
        if( vol_np*( 1+vol_ncp ) > RKFD_WAVE ) FAIL( "Volume plugin: %d pairs x ( 1 + %d conditions ) exceed 64 constraints", vol_np, vol_ncp );
        if( vol_pv < 10 ) vol_pv = 10;
        /* (a face loop of n vertices clipped by the k planes of the other shape can reach n + k vertices - the bound above - but a
         *  cylinder's 16-gon against a 34-face cylinder is 50 vertices x 64 lanes x 24 bytes = 77 KB of LDS for polygons that in
         *  practice gain a handful: 32 vertices per lane, a polygon beyond that is reported at run time - status 2) */
        if( vol_pv > 32 && maxloop + 8 <= 32 ) vol_pv = 32;
        vol_nf = maxfaces;
      }
    }
  }
  if( vol_np > 0 ) max_rigid = 2*vol_np;      /* six rows per pair in the arrays the MLCP phase sizes by 3*max_rigid */
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
    std::vector<std::vector<int> > iters;          /* iteration -> NG slots -> link */
    std::vector<int> level_first_iter( nlevel+1, 0 );
    if( NG == 8 ){
    /* assign slots bottom-up */
      std::vector<std::vector<std::vector<int> > > per_level( nlevel );
      for( int d=nlevel-1; d>=0; d-- ){
        const int n = level_off[d+1] - level_off[d];
        const int nch = ( n + NG-1 ) / NG;
        per_level[d].assign( nch, std::vector<int>( NG, -1 ) );
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
          while( per_level[d][c][sidx] >= 0 ){ sidx++; if( sidx == NG ){ sidx = 0; c++; } }
          per_level[d][c][sidx] = rest[k]; slot[rest[k]] = sidx;
        }
      }
      for( int d=0; d<nlevel; d++ )
        for( size_t c=0; c<per_level[d].size(); c++ ){
          for( int g=0; g<NG; g++ ) if( per_level[d][c][g] >= 0 ) iter[per_level[d][c][g]] = nsched;
          iters.push_back( per_level[d][c] );
          nsched++;
        }
    } else {
      /* Fewer lane groups than the widest level has links (two instances per wavefront: four groups): a level-by-level schedule
       * would split such levels over several iterations and break the register hand-offs along chains (a child's Ia stays in its
       * group's registers only when its parent follows in the very next iteration, in the same group) - on the humanoid 22 of 25
       * links would then stage their Ia in LDS (4.9 KB more per instance: 8 instead of 10 instances per CU).  The sweeps need no
       * levels, only a topological order (a parent after all its children in sweep 2, before them in sweep 3), so the links are
       * LIST-scheduled here: every group follows a chain as far as it goes - the link it processed last hands over to one of its
       * children in the next iteration -, free groups take the ready link with the longest chain below it. */
      std::vector<int> height( NL, 1 );
      for( int i=NL-1; i>=0; i-- ) if( R_parent[i] >= 0 && height[i]+1 > height[R_parent[i]] ) height[R_parent[i]] = height[i]+1;
      std::vector<int> prev( NG, -1 );
      int left = NL;
      while( left > 0 ){
        std::vector<int> cur( NG, -1 );
        /* continue the chains */
        for( int g=0; g<NG; g++ ){
          const int p = prev[g];
          if( p < 0 ) continue;
          int best = -1;
          for( int c=child_off[p]; c<child_off[p+1]; c++ ){
            const int ch = child_idx[c];
            if( iter[ch] < 0 && ( best < 0 || height[ch] > height[best] ) ) best = ch;
          }
          if( best >= 0 ){ cur[g] = best; iter[best] = nsched; slot[best] = g; left--; }
        }
        /* free groups: ready links (parent done in an EARLIER iteration, or no parent), longest chain first */
        for( int g=0; g<NG; g++ ){
          if( cur[g] >= 0 ) continue;
          int best = -1;
          for( int i=0; i<NL; i++ ){
            if( iter[i] >= 0 ) continue;
            const int p = R_parent[i];
            if( p >= 0 && ( iter[p] < 0 || iter[p] >= nsched ) ) continue;
            if( best < 0 || height[i] > height[best] ) best = i;
          }
          if( best < 0 ) break;
          cur[g] = best; iter[best] = nsched; slot[best] = g; left--;
        }
        iters.push_back( cur );
        prev = cur;
        nsched++;
      }
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
          if( iter[ch] == iter[i]+1 && slot[ch] == slot[i] && R_jtype[ch] != RKFD_JOINT_FLOAT ) carried[ch] = 1;
        }
      }
      for( int i=0; i<NL; i++ ){
        pslot[i] = -1;
        if( R_parent[i] >= 0 && ( R_jtype[i] != RKFD_JOINT_FLOAT || R_brf[i] ) && !carried[i] ) pslot[i] = npool++;
        fslot[i] = -1;
        if( R_jtype[i] == RKFD_JOINT_FLOAT ) fslot[i] = nfloat++;
      }
    }
    for( int t=-2; t<nsched+2; t++ )
      for( int g=0; g<NG; g++ ){
        int rec[4] = { -1, 0, 0, 0 };
        const int i = ( t >= 0 && t < nsched ) ? iters[t][g] : -1;
        if( i >= 0 ){
          const int nchild = child_off[i+1] - child_off[i];
          int flags = 0;
          if( nchild == 1 ){
            const int ch = child_idx[child_off[i]];
            if( iter[ch] == t+1 && slot[ch] == g && R_jtype[ch] != RKFD_JOINT_FLOAT ) flags |= 1;
          }
          if( R_parent[i] >= 0 && iter[R_parent[i]] == t-1 && slot[R_parent[i]] == g ) flags |= 2;
          if( nchild > 255 || pslot[i]+1 > 255 || fslot[i]+1 > 255 ) FAIL( "schedule record overflow" );
          rec[0] = i; rec[1] = linfo[i];
          rec[2] = nchild | ( flags << 8 ) | ( ( pslot[i]+1 ) << 16 ) | ( ( fslot[i]+1 ) << 24 );
          rec[3] = child_off[i];
        }
        sched.insert( sched.end(), rec, rec+4 );
      }
  }

  /* rows nlevel+1, nlevel+2 of the path table: float slot of a link (its CHOL / XF slot), link of a float slot */
  for( int i=0; i<NL; i++ ){
    pathlink[(size_t)NL*( nlevel+1 )+i] = fslot[i];
    if( fslot[i] >= 0 ) pathlink[(size_t)NL*( nlevel+2 )+fslot[i]] = i;
  }

  Blob b;
  rkfdDevModel dm;
  memset( &dm, 0, sizeof(dm) );
  dm.nlink = NL; dm.nlink_model = NLm; dm.ndof = ND; dm.ncand = NC; dm.nlevel = nlevel; dm.nround = nround; dm.nci = m->nci;
  dm.solver = m->solver; dm.max_iter = m->max_iter; dm.maxrg = max_rigid;
  dm.dt = m->dt; dm.fric_w = m->friction_weight;
  dm.nsched = nsched; dm.npool = npool; dm.nfloat = nfloat; dm.ngroup = NG;
  dm.has_brf = has_brf;
  dm.anchor = -1;
  for( int i=0; i<NL; i++ ) if( !is_static[i] ){ dm.anchor = i; break; }
  /* contact capacities */
  /* active-contact slots: up to 16 when elastic (penalty) contacts can occur, and at least the
   * rigid capacity when rigid ones can */
  bool has_elastic = false, has_rigid = false;
  for( int j=0; j<NC; j++ ){
    if( m->ci_type[cci[j]] == RKFD_CONTACT_RIGID ) has_rigid = true; else has_elastic = true;
  }
  dm.pyramid = m->pyramid > 0 ? m->pyramid : 8;
  dm.vert_rigid = ( m->solver == RKFD_SOLVER_VERT && has_rigid && max_rigid > 0 ) ? 1 : 0;
  /* 2: at most 24 unknowns - the factor of the QP's Q lives in registers (rkfd_dev_vertqp.h: rkfdQpFactor), W = L^-1 C' in the
   * storage of the contact matrix and the Schur complement in that of the factor: no separate W in LDS */
  if( dm.vert_rigid && 3*max_rigid <= RKFD_QP_NQ_MAX ) dm.vert_rigid = 2;
  /* 3: more unknowns or more pyramid faces than lanes - the QP's wide form, every vector in LDS and every loop strided by the
   * wavefront (rkfd_vert_qp_wide) */
  if( dm.vert_rigid && ( 3*max_rigid > RKFD_WAVE || dm.pyramid*max_rigid > RKFD_WAVE ) ) dm.vert_rigid = 3;
  dm.qscr_alias = 0;      /* (the QP's reductions go through registers since round 3: no scratch) */
  dm.ma_packed = 0;
  dm.ma_size = dm.vert_rigid ? 3*max_rigid*( 3*max_rigid+1 ) : 9*max_rigid*max_rigid;   /* full rows (odd stride while a slot is free / for the Vert QP); see ma_packed below */
  if( dm.vert_rigid && (size_t)dm.pyramid*max_rigid > 3*RKFD_WAVE )
    FAIL( "Vert plugin: pyramid faces x rigid contact capacity exceeds 192 (the active set is kept as three 64-bit words)" );
  int maxact = has_elastic ? ( NC < 16 ? NC : 16 ) : 0;
  if( has_rigid && max_rigid > maxact ) maxact = max_rigid < NC ? max_rigid : NC;
  if( NC > 0 && maxact < 1 ) maxact = 1;
  if( vol_np > maxact ) maxact = vol_np;      /* (the moving sides of the pairs go where those of the contact vertices go) */
  int nside = 1;
  for( int j=0; j<NC; j++ )
    if( m->ci_type[cci[j]] == RKFD_CONTACT_RIGID && !is_static[cA[j]] && !is_static[cB[j]] ) nside = 2;
  for( int k=0; k<vol_npair; k++ ) if( !is_static[vol_pair[8*k]] && !is_static[vol_pair[8*k+1]] ) nside = 2;
  dm.vol_npair = vol_np > 0 ? vol_npair : 0; dm.vol_np = vol_np; dm.vol_ncp = vol_ncp; dm.vol_pv = vol_pv; dm.vol_nf = vol_nf;
  if( vol_np > 0 ){ dm.vert_rigid = 0; dm.qscr_alias = 0; dm.ma_packed = 0; dm.ma_size = 6*vol_np*( 6*vol_np+1 ); }
  dm.maxact = maxact; dm.nside = nside;
  /* kernel variants: bit 2 the Vert QP's Q = A'A on the matrix cores (on: +5 % on config 4 under the Vert plugin), bit 3 the
   * grouped Gauss-Seidel off, bit 5 its sweep-order storage off - the last two exist for the tests that show the grouped form
   * reproduces the one-after-the-other loop bit for bit (rkfdDebugVariants; nothing is read from the environment) */
  dm.mlcp_mfma = 4 ^ g_debug_variants;
  const size_t Mrows = 3*(size_t)max_rigid;
  /* probe scratch: one row per tree level plus six for a float root, per side; it overlays the
   * C|PA block of the link arrays (dead while the contact problem is solved) when it fits */
  int pu_d0 = nlevel;
  for( int i=0; i<NL; i++ ) if( ( RKFD_JT_IS1( R_jtype[i] ) || R_brf[i] ) && depth[i] < pu_d0 ) pu_d0 = depth[i];      /* (an unbroken breakable float joint is a level a probe path passes) */
  if( pu_d0 == nlevel ) pu_d0 = nlevel > 0 ? nlevel-1 : 0;      /* (no 1-DoF joint at all: one unused row) */
  const int npurow = nlevel - pu_d0 + ( nfloat > 0 ? 6 : 0 );
  dm.npurow = npurow; dm.pu_d0 = pu_d0;
  dm.pu_alias = ( (size_t)nside*npurow*Mrows <= (size_t)12*NL ) ? 1 : 0;
  std::vector<int> dofkind( ND ? ND : 1, 0 );
  for( int i=0; i<NL; i++ ){
    if( R_jtype[i] == RKFD_JOINT_FLOAT ){ dofkind[R_dofoff[i]+3] = 1; dofkind[R_dofoff[i]+4] = 2; dofkind[R_dofoff[i]+5] = 2; }
    if( R_jtype[i] == RKFD_DJT_SPHX ){ dofkind[R_dofoff[i]] = 1; dofkind[R_dofoff[i]+1] = 2; dofkind[R_dofoff[i]+2] = 2; }   /* angle-axis coordinates */
  }
  if( nround > RKFD_MAX_ROUND ) FAIL( "tree too deep" );
  /* record offsets first (the vector may reallocate), then resolve */
  struct Ent { const void **slot; size_t off; };
  std::vector<Ent> ents;
#define PUT(field, src, bytes) ents.push_back( Ent{ (const void **)&dm.field, b.put( src, bytes ) } )
  PUT( parent, R_parent.data(), sizeof(int)*NL ); PUT( jtype, R_jtype.data(), sizeof(int)*NL );
  PUT( dofoff, R_dofoff.data(), sizeof(int)*NL ); PUT( mtype, R_mtype.data(), sizeof(int)*NL );
  PUT( orig, orig.data(), sizeof(int)*NL );
  PUT( brf, R_brf.data(), sizeof(int)*NL ); PUT( brk_f, R_brk_f.data(), sizeof(double)*NL ); PUT( brk_t, R_brk_t.data(), sizeof(double)*NL );
  PUT( dofkind, dofkind.data(), sizeof(int)*( ND ? ND : 1 ) );
  PUT( depth, depth.data(), sizeof(int)*NL ); PUT( is_static, is_static.data(), sizeof(int)*NL );
  PUT( org, R_org.data(), sizeof(double)*12*NL ); PUT( mass, R_mass.data(), sizeof(double)*NL );
  PUT( com, R_com.data(), sizeof(double)*3*NL ); PUT( inertia, R_inertia.data(), sizeof(double)*9*NL );
  PUT( stiff, R_stiff.data(), sizeof(double)*NL ); PUT( visc, R_visc.data(), sizeof(double)*NL );
  PUT( coulomb, R_coulomb.data(), sizeof(double)*NL ); PUT( sfric, R_sfric.data(), sizeof(double)*NL );
  PUT( mot_k, R_mot_k.data(), sizeof(double)*NL ); PUT( mot_admit, R_mot_admit.data(), sizeof(double)*NL );
  PUT( mot_vmax, R_mot_vmax.data(), sizeof(double)*NL ); PUT( mot_vmin, R_mot_vmin.data(), sizeof(double)*NL );
  PUT( mot_gear, R_mot_gear.data(), sizeof(double)*NL ); PUT( mot_inertia, R_mot_inertia.data(), sizeof(double)*NL );
  PUT( anc, anc.data(), sizeof(int)*anc.size() );
  PUT( level_off, level_off.data(), sizeof(int)*( nlevel+1 ) ); PUT( level_link, level_link.data(), sizeof(int)*NL );
  PUT( child_off, child_off.data(), sizeof(int)*( NL+1 ) ); PUT( child_idx, child_idx.data(), sizeof(int)*NL );
  PUT( pathlink, pathlink.data(), sizeof(int)*pathlink.size() );
  PUT( linfo, linfo.data(), sizeof(int)*NL ); PUT( cinfo, cinfo.data(), sizeof(int)*NC );
  PUT( sched, sched.data(), sizeof(int)*sched.size() );
  PUT( pslot, pslot.data(), sizeof(int)*NL );
  PUT( cand_linkA, cA.data(), sizeof(int)*NC ); PUT( cand_linkB, cB.data(), sizeof(int)*NC );
  PUT( cand_foff, cfo.data(), sizeof(int)*NC ); PUT( cand_nf, cnf.data(), sizeof(int)*NC );
  PUT( cand_ci, cci.data(), sizeof(int)*NC ); PUT( cand_vert, cv.data(), sizeof(double)*3*NC ); PUT( cand_bs, cbs.data(), sizeof(double)*4*( NC ? NC : 1 ) );
  dm.has_slide = has_slide;
  PUT( cs_mode, csm.data(), sizeof(int)*csm.size() ); PUT( cs_par, csp.data(), sizeof(double)*csp.size() );
  PUT( planes, planes_d.data(), sizeof(double)*4*nplane );
  if( vol_pair.empty() ) vol_pair.assign( 8, 0 );
  if( vol_loop.empty() ) vol_loop.assign( 2, 0 );
  if( vol_lplane.empty() ) vol_lplane.assign( 4, 0.0 );
  if( vol_lvert.empty() ) vol_lvert.assign( 3, 0.0 );
  if( vol_slide.empty() ) vol_slide.assign( 16, 0.0 );
  PUT( vol_pair, vol_pair.data(), sizeof(int)*vol_pair.size() ); PUT( vol_loop, vol_loop.data(), sizeof(int)*vol_loop.size() );
  PUT( vol_lplane, vol_lplane.data(), sizeof(double)*vol_lplane.size() ); PUT( vol_lvert, vol_lvert.data(), sizeof(double)*vol_lvert.size() );
  PUT( vol_slide, vol_slide.data(), sizeof(double)*vol_slide.size() );
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
  {
    /* host-only table, appended to the blob's allocation lifetime through a separate copy */
    double *rf = (double *)malloc( sizeof(double)*refT.size() );
    if( !rf ) FAIL( "out of memory" );
    memcpy( rf, refT.data(), sizeof(double)*refT.size() );
    out->ref_frame = rf; out->ncand = NC;
  }
  /* LDS bytes one instance needs (must match rkfd_lds_carve in rkfd_device.h).  Two passes: with the contact matrix
   * as full rows, and - PGS worlds only - as a packed lower triangle, which is taken when it lets one more instance
   * share a CU (LDS is handed out in 1280-byte pieces, 128 per CU) */
  size_t lds_full = 0;
  for( int pass=0; pass<2; pass++ ){
    if( pass == 1 ){
      const size_t Mr = 3*(size_t)max_rigid;
      if( dm.vert_rigid || max_rigid <= 0 || vol_np > 0 ) break;
      /* (one instance per wavefront: the Gauss-Seidel form that broadcasts through the DPP operand exists for full rows only - its
       *  index arithmetic on the packed triangle costs registers that kernel does not have - so up to 16 contacts a packed matrix
       *  means the general loop whenever the count is not 4 or 8: the 26-DoF humanoid 10.2 M steps/s with full rows at ten per CU,
       *  8.8 M packed at eleven.  The kernel with two instances per wavefront has the registers and takes the packed form.) */
      if( NG == 8 && max_rigid <= 16 ) break;
      dm.ma_packed = 1; dm.ma_size = (int)( Mr*( Mr+1 )/2 );
    }
    const size_t M = 3*(size_t)max_rigid;
    const size_t pool = (size_t)36*npool > (size_t)6*NL ? (size_t)36*npool : (size_t)6*NL;   /* Ia pool | second half of the world frames */
    size_t stage = (size_t)14*NL + pool;                         /* inertia staging + Ia pool ...   */
    if( (size_t)dm.ma_size > stage ) stage = (size_t)dm.ma_size;   /* ... shared with the contact matrix */
    const size_t dbl = (size_t)NL*( 5*6 + 3 ) + stage + (size_t)33*nfloat
                     + (size_t)maxact*( 18 + ( NC > RKFD_WAVE/2 ? 3 : 0 ) + ( dm.has_slide ? 6 : 0 ) ) + ( dm.vert_rigid ? 2 : 1 )*M + ( dm.pu_alias ? 0 : (size_t)nside*npurow*M )
                     + ( dm.vert_rigid ? ( dm.vert_rigid == 2 ? 0 : M*M ) + M*( M+1 )/2 + 5*M + 3*M : 0 )   /* Vert QP: QL, QW, QV, CR */
                     + ( dm.vert_rigid == 3 ? (size_t)dm.pyramid*M + (size_t)dm.pyramid*max_rigid : 0 )             /* ... wide form: QG, QY */
                     + ( vol_np > 0 ? (size_t)RKFD_VOL_LDS_DOUBLES( vol_np, vol_ncp, vol_pv, vol_nf, dm.pyramid ) : 0 );
    /* two instances per wavefront: the world's static tables - CIp, LI (ints), CHP, CFO (16 bit), PL (bytes) - once per wavefront
     * (rkfdDevModel.lds_shared); not with breakable joints, whose link info and path tops are per instance */
    const bool shr = NG == 4 && !has_brf;
    const size_t sh_ints = (size_t)NC + (size_t)NL, sh_bytes = (size_t)2*NL + (size_t)2*NC + ( max_rigid > 0 ? (size_t)NL*( nlevel+3 ) : 0 );
    dm.lds_shared = shr ? (int)( ( sh_ints*sizeof(int) + sh_bytes + 15 ) & ~(size_t)15 ) : 0;
    const size_t ints = (size_t)NC + (size_t)nside*maxact + ( vol_np > 0 ? 12 + 2*vol_np : ( NC > 0 ? 8 : 4 ) ) + (size_t)NL     /* CIp, tgt, cnt (VI), LI */
                      + ( RKFD_GC_NEEDED( (int)M ) ? RKFD_GC_INTS : 0 )                                                             /* GC */
                      - ( shr ? sh_ints : 0 );
    const size_t bytes = (size_t)2*NL + (size_t)5*NC + (size_t)5*maxact - ( shr ? sh_bytes : 0 ) + ( dm.vert_rigid ? M : 0 ) + ( dm.vert_rigid == 3 ? (size_t)( dm.pyramid+1 )*max_rigid : 0 )                                     /* CHP (16 bit), act typ asl (bytes) */
                       + ( max_rigid > 0 ? (size_t)NL*( nlevel+3 ) : 0 )                 /* PL */
                       + ( has_brf ? (size_t)NL : 0 );                                   /* BRK */
    out->lds_bytes = dbl*sizeof(double) + ints*sizeof(int) + bytes;
    out->lds_bytes = ( out->lds_bytes + 15 ) & ~(size_t)15;
    /* (what a workgroup asks for: one instance, or two and the shared tables) */
    const size_t wg = NG == 4 ? 2*out->lds_bytes + (size_t)dm.lds_shared : out->lds_bytes;
    const size_t wg_full = NG == 4 ? 2*lds_full + (size_t)dm.lds_shared : lds_full;
    if( pass == 0 ) lds_full = out->lds_bytes;
    else if( 128/( ( wg+1279 )/1280 ) <= 128/( ( wg_full+1279 )/1280 ) ){
      /* no instance gained: stay with full rows */
      dm.ma_packed = 0; dm.ma_size = 9*max_rigid*max_rigid; out->lds_bytes = lds_full;
    }
    if( getenv( "RKFD_DEVMODEL_DUMP" ) )
      fprintf( stderr, "rkfd devmodel: NL %d ND %d NC %d nlevel %d npool %d nfloat %d maxact %d nside %d npurow %d pu_alias %d M %d stage %zu (staging %zu) vert %d -> %zu B of LDS\n",
               NL, ND, NC, nlevel, npool, nfloat, maxact, nside, npurow, dm.pu_alias, (int)M, stage, (size_t)14*NL + pool, dm.vert_rigid, out->lds_bytes );
  }
  out->dm.ma_packed = dm.ma_packed; out->dm.ma_size = dm.ma_size;
  out->dm.lds_instance = (int)out->lds_bytes; out->dm.lds_shared = dm.lds_shared;
  if( NG == 4 ){
    /* two instances per wavefront: everything that is one lane per item must fit the 32 lanes of an instance */
    if( NL > 32 || ND > 32 || dm.maxact > 32 || max_rigid > 16 || dm.vert_rigid || vol_np > 0 )
      FAIL( "two instances per wavefront need a world of at most 32 links, 32 joint coordinates, 32 contact slots and 16 rigid contact vertices, without the Vert QP / the Volume plugin (this one: %d links, %d coordinates, %d slots, max_rigid %d)", NL, ND, dm.maxact, max_rigid );
  }
  { const char *e = getenv( "RKFD_DEBUG_POISON_LDS" ); out->dm.lds_poison = ( e && atoi( e ) > 0 ) ? (int)( out->lds_bytes/4 ) : 0; }
  return 0;
#undef FAIL
}

extern "C" void rkfd_devmodel_free(rkfdDevModelHost *h)
{
  if( h ){ free( h->blob ); h->blob = NULL; free( (void *)h->ref_frame ); h->ref_frame = NULL; }
}

extern "C" void rkfd_devmodel_rebase(rkfdDevModel *dm, const void *from, const void *to)
{
  const ptrdiff_t d = (const char *)to - (const char *)from;
#define RB(f) dm->f = (decltype(dm->f))( (const char *)dm->f + d )
  RB(parent); RB(jtype); RB(dofoff); RB(mtype); RB(depth); RB(is_static);
  RB(org); RB(mass); RB(com); RB(inertia); RB(stiff); RB(visc); RB(coulomb); RB(sfric);
  RB(mot_k); RB(mot_admit); RB(mot_vmax); RB(mot_vmin); RB(mot_gear); RB(mot_inertia);
  RB(anc); RB(level_off); RB(level_link); RB(child_off); RB(child_idx); RB(pathlink); RB(linfo); RB(sched); RB(cinfo); RB(pslot); RB(orig); RB(dofkind); RB(brf); RB(brk_f); RB(brk_t);
  RB(cand_linkA); RB(cand_linkB); RB(cand_foff); RB(cand_nf); RB(cand_ci); RB(cand_vert); RB(cand_bs); RB(cs_mode); RB(cs_par); RB(planes);
  RB(vol_pair); RB(vol_loop); RB(vol_lplane); RB(vol_lvert); RB(vol_slide);
  RB(ci_type); RB(ci_sf); RB(ci_kf); RB(ci_k); RB(ci_l); RB(ci_e); RB(ci_v);
#undef RB
}

extern "C" void rkfd_ref_to_device(const rkfdDevModelHost *h, double *ref, size_t n)
{
  for( size_t k=0; k<n; k++ ){
    const double *T = &h->ref_frame[12*( k % (size_t)h->ncand )];
    double *r = &ref[3*k], o[3];
    for( int a=0; a<3; a++ ) o[a] = T[3*a]*r[0] + T[3*a+1]*r[1] + T[3*a+2]*r[2] + T[9+a];
    r[0] = o[0]; r[1] = o[1]; r[2] = o[2];
  }
}
extern "C" void rkfd_ref_to_model(const rkfdDevModelHost *h, double *ref, size_t n)
{
  for( size_t k=0; k<n; k++ ){
    const double *T = &h->ref_frame[12*( k % (size_t)h->ncand )];
    double *r = &ref[3*k], d[3] = { r[0]-T[9], r[1]-T[10], r[2]-T[11] };
    for( int a=0; a<3; a++ ) r[a] = T[a]*d[0] + T[3+a]*d[1] + T[6+a]*d[2];
  }
}
