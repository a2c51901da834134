/* rkfd_oracle_volume.h - the Volume plugin's rigid branch (reference src/rkfd_volume.c), part of the CPU oracle
 * (TEST INFRASTRUCTURE ONLY, PARITY UNPINNED - see rkfd_oracle.h).  Included by rkfd_oracle.c, not a public header.
 *
 * The reference file is restated function by function (line numbers cited).  What it delegates to un-vendored
 * libraries is restated from their published algorithms and tagged [UNVERIFIED-DEP]:
 *   rkCDColVolBREPVert (RoKi rk_cd; reference src/rkfd_volume.c:1007) - per colliding pair the intersection volume of
 *     the two shapes (zPH3D colvol), its barycentre (center), the contact normal (norm) and the frame axis[3].  Here:
 *     a pair collides when a vertex of one shape lies inside the other ("Vert"); both shapes are convex, so the
 *     intersection is the set of face polygons of either shape clipped by the planes of the other (Sutherland-Hodgman),
 *     fan-triangulated; center = centroid of that volume; norm = direction of the summed area vectors of the faces
 *     the volume takes from cell[1] (equal, the surface being closed, to minus the sum over the faces from cell[0]):
 *     the direction that pushes cell[0] out of cell[1]; axis = ( norm, the orthonormal complement as in the vertex
 *     path ).  The contact-plane list is rebuilt on every evaluation.
 *   zLPSolveSimplex / zLPFeasibleBase (ZM zm_opt; :685,837,839) - min c'x s.t. Ax = b, x >= 0 by the two-phase
 *     tableau simplex method (entering column: most negative reduced cost, Bland's rule after 64 pivots; leaving row:
 *     minimum ratio, lowest basic index among equals); false when infeasible or unbounded.
 *   zList order - zListInsertHead appends at the end zListForEach reaches last; rkCDPlaneListQuickSort leaves the
 *     list ascending in the order zListForEach visits it.
 *   zMat6D - e[row block][column block]; _zVec3DOuterProdToMat3D(p) = [p x], _zVec3DTripleProdToMat3D(a,b) = [a x][b x].
 */
#ifndef RKFD_ORACLE_VOLUME_H
#define RKFD_ORACLE_VOLUME_H

/* ------------------------------------------------------------------------ */
/* face loops of the convex shapes (link frame): for every plane of a shape the vertices lying on it, counter-clockwise
 * seen from outside.  Coplanar duplicates of an earlier plane get an empty loop. */
static void vol_prepare(rkfdOracle *o)
{
  const rkfdModel *m = o->m;
  int s, f, f2, v, i, j, np = m->shape_foff[m->nshape], cap = 0, n = 0;
  if( o->vol_ready ) return;
  o->fl_off = (int *)zalloc( sizeof(int)*( np+1 ) );
  for( s=0; s<m->nshape; s++ ) cap += ( m->shape_foff[s+1] - m->shape_foff[s] )*( m->shape_voff[s+1] - m->shape_voff[s] );
  o->fl_idx = (int *)zalloc( sizeof(int)*( cap+1 ) );
  for( s=0; s<m->nshape; s++ )
    for( f=m->shape_foff[s]; f<m->shape_foff[s+1]; f++ ){
      const double *pl = &m->planes[4*f];
      double c[3] = {0,0,0}, t1[3], t2[3], ang[VOL_MAXPV]; int idx[VOL_MAXPV], k = 0, dup = 0;
      o->fl_off[f] = n;
      for( f2=m->shape_foff[s]; f2<f; f2++ ){
        const double *p2 = &m->planes[4*f2];
        if( fabs( pl[0]-p2[0] ) < 1e-9 && fabs( pl[1]-p2[1] ) < 1e-9 && fabs( pl[2]-p2[2] ) < 1e-9 && fabs( pl[3]-p2[3] ) < 1e-9 ) dup = 1;
      }
      if( dup ) continue;
      for( v=m->shape_voff[s]; v<m->shape_voff[s+1] && k<VOL_MAXPV; v++ )
        if( fabs( v3_dot( pl, &m->verts[3*v] ) - pl[3] ) < 1e-9 ){
          int same = 0;
          for( i=0; i<k; i++ ){
            const double *a = &m->verts[3*idx[i]], *b = &m->verts[3*v];
            if( fabs( a[0]-b[0] ) < 1e-12 && fabs( a[1]-b[1] ) < 1e-12 && fabs( a[2]-b[2] ) < 1e-12 ) same = 1;
          }
          if( !same ) idx[k++] = v;
        }
      if( k < 3 ) continue;
      for( i=0; i<k; i++ ) v3_cat( c, 1.0/k, &m->verts[3*idx[i]] );
      ortho_space( pl, t1, t2 );
      for( i=0; i<k; i++ ){
        double d[3];
        v3_sub( &m->verts[3*idx[i]], c, d );
        ang[i] = atan2( v3_dot( d, t2 ), v3_dot( d, t1 ) );
      }
      for( i=1; i<k; i++ ){            /* insertion sort by angle */
        double a = ang[i]; int id = idx[i];
        for( j=i-1; j>=0 && ang[j]>a; j-- ){ ang[j+1] = ang[j]; idx[j+1] = idx[j]; }
        ang[j+1] = a; idx[j+1] = id;
      }
      for( i=0; i<k; i++ ) o->fl_idx[n++] = idx[i];
    }
  o->fl_off[np] = n;
  o->vp = (VolPair *)zalloc( sizeof(VolPair)*( m->npair ? m->npair : 1 ) );
  o->vp_type = (int *)zalloc( sizeof(int)*( m->npair ? m->npair : 1 ) );
  o->vol_ready = 1;
}

/* one face polygon against one half space n.x - d <= 0 (Sutherland-Hodgman) */
static int vol_clip(double (*p)[3], int n, const double *pl, double d)
{
  double q[VOL_MAXPV][3], s[VOL_MAXPV];
  int i, k = 0;
  for( i=0; i<n; i++ ) s[i] = v3_dot( pl, p[i] ) - d;
  for( i=0; i<n; i++ ){
    int j = i+1 == n ? 0 : i+1;
    if( s[i] <= 0 && k < VOL_MAXPV ){ v3_copy( p[i], q[k] ); k++; }
    if( ( ( s[i] < 0 && s[j] > 0 ) || ( s[i] > 0 && s[j] < 0 ) ) && k < VOL_MAXPV ){
      const double t = s[i]/( s[i] - s[j] );
      q[k][0] = p[i][0] + t*( p[j][0]-p[i][0] ); q[k][1] = p[i][1] + t*( p[j][1]-p[i][1] ); q[k][2] = p[i][2] + t*( p[j][2]-p[i][2] );
      k++;
    }
  }
  for( i=0; i<k; i++ ) v3_copy( q[i], p[i] );
  return k;
}

static void vol_push_tri(VolPair *vp, const double *a, const double *b, const double *c, const double *nw)
{
  double *t;
  if( vp->ntri == vp->captri ){
    vp->captri = vp->captri ? 2*vp->captri : 64;
    vp->tri = (double *)realloc( vp->tri, sizeof(double)*12*vp->captri );
  }
  t = &vp->tri[12*vp->ntri++];
  v3_copy( a, t ); v3_copy( b, t+3 ); v3_copy( c, t+6 ); v3_copy( nw, t+9 );
}

/* faces of shape sa inside shape sb -> triangles of the intersection volume; the summed area vector of what was kept */
static void vol_clip_shape(rkfdOracle *o, VolPair *vp, int sa, int sb, double *asum)
{
  const rkfdModel *m = o->m;
  const Link *A = &o->lk[m->shape_link[sa]], *B = &o->lk[m->shape_link[sb]];
  int f, g, i;
  v3_zero( asum );
  for( f=m->shape_foff[sa]; f<m->shape_foff[sa+1]; f++ ){
    double p[VOL_MAXPV][3], nw[3], av[3] = {0,0,0};
    int n = o->fl_off[f+1] - o->fl_off[f];
    if( n < 3 ) continue;
    for( i=0; i<n; i++ ){ m3_mulv( A->R, &m->verts[3*o->fl_idx[o->fl_off[f]+i]], p[i] ); v3_add( p[i], A->p, p[i] ); }
    m3_mulv( A->R, &m->planes[4*f], nw );
    for( g=m->shape_foff[sb]; g<m->shape_foff[sb+1] && n>=3; g++ ){
      double gw[3];
      if( o->fl_off[g+1] - o->fl_off[g] < 3 ) continue;       /* duplicate of an earlier plane */
      m3_mulv( B->R, &m->planes[4*g], gw );
      n = vol_clip( p, n, gw, m->planes[4*g+3] + v3_dot( gw, B->p ) );
    }
    if( n < 3 ) continue;
    for( i=1; i+1<n; i++ ){
      double e1[3], e2[3], x[3];
      v3_sub( p[i], p[0], e1 ); v3_sub( p[i+1], p[0], e2 ); v3_cross( e1, e2, x );
      if( v3_norm( x ) < 1e-24 ) continue;
      v3_cat( av, 0.5, x );
      vol_push_tri( vp, p[0], p[i], p[i+1], nw );
    }
    v3_add( asum, av, asum );
  }
}

static int vol_any_vertex_inside(const rkfdOracle *o, int sa, int sb)
{
  const rkfdModel *m = o->m;
  const Link *A = &o->lk[m->shape_link[sa]], *B = &o->lk[m->shape_link[sb]];
  int v, f;
  for( v=m->shape_voff[sa]; v<m->shape_voff[sa+1]; v++ ){
    double x[3], r[3], y[3], smax = -HUGE_VAL;
    m3_mulv( A->R, &m->verts[3*v], x ); v3_add( x, A->p, x );
    v3_sub( x, B->p, r ); m3_tmulv( B->R, r, y );
    for( f=m->shape_foff[sb]; f<m->shape_foff[sb+1]; f++ ){
      double s = v3_dot( &m->planes[4*f], y ) - m->planes[4*f+3];
      if( s > smax ) smax = s;
    }
    if( smax < TOL ) return 1;
  }
  return 0;
}

/* rkCDColVolBREPVert [UNVERIFIED-DEP, see the header of this file] followed by rkFDCDUpdate (reference src/rkfd_cd.c:33-49)
 * for the rigid pairs */
static void vol_collision(rkfdOracle *o)
{
  const rkfdModel *m = o->m;
  int pr, i, k;
  vol_prepare( o );
  o->nvp = 0;
  for( pr=0; pr<m->npair; pr++ ){
    VolPair *vp = &o->vp[o->nvp];
    int sa = m->pair_shape[2*pr], sb = m->pair_shape[2*pr+1];
    double asA[3], asB[3], ref[3], v6 = 0, cen[3] = {0,0,0};
    if( m->ci_type[m->pair_ci[pr]] != RKFD_CONTACT_RIGID ) continue;
    if( !vol_any_vertex_inside( o, sa, sb ) && !vol_any_vertex_inside( o, sb, sa ) ) continue;
    if( o->vol_raw[sa] || o->vol_raw[sb] ){ o->vol_guard_hits++; continue; }      /* a guarded pair (rkfdOracleCreate) */
    vp->pair = pr; vp->ci = m->pair_ci[pr]; vp->sa = sa; vp->sb = sb;
    vp->la = m->shape_link[sa]; vp->lb = m->shape_link[sb];
    vp->ntri = 0; vp->ncp = 0;
    for( k=0; k<6; k++ ) vp->wrench[k] = 0;
    vol_clip_shape( o, vp, sa, sb, asA );
    vol_clip_shape( o, vp, sb, sa, asB );
    if( vp->ntri < 4 ) continue;
    /* volume and barycentre (zPH3DBarycenter): signed tetrahedra over a reference point on the surface */
    v3_copy( vp->tri, ref );
    for( i=0; i<vp->ntri; i++ ){
      const double *t = &vp->tri[12*i];
      double a[3], b[3], c[3], x[3], w;
      v3_sub( t, ref, a ); v3_sub( t+3, ref, b ); v3_sub( t+6, ref, c );
      v3_cross( b, c, x ); w = v3_dot( a, x );
      v6 += w;
      for( k=0; k<3; k++ ) cen[k] += w*( a[k] + b[k] + c[k] );
    }
    if( !( v6 > 1e-30 ) ) continue;
    for( k=0; k<3; k++ ) vp->center[k] = ref[k] + cen[k]/( 4.0*v6 );
    vp->volume = v6/6.0;
    if( !( v3_norm( asB ) > 1e-30 ) ) continue;          /* cell[0] wholly inside cell[1]: no direction to push along */
    v3_mul( asB, 1.0/v3_norm( asB ), vp->norm );
    v3_copy( vp->norm, vp->axis );
    ortho_space( vp->norm, vp->axis+3, vp->axis+6 );
    o->nvp++;
  }
}

/* ------------------------------------------------------------------------ */
/* 6-D point kinematics: rkFDLinkPointWldVel6D / rkFDChainPointRelativeVel6D (reference src/rkfd_util.c:62-89; no slide
 * mode in this plugin, :83-85), rkFDLinkPointWldAcc6D / rkFDChainPointRelativeAcc6D (:120-145) */
static void vol_rel_vel6(const rkfdOracle *o, const VolPair *vp, double *v)
{
  double a[3], b[3], wa[3], wb[3];
  link_point_vel( &o->lk[vp->la], vp->center, a ); link_point_vel( &o->lk[vp->lb], vp->center, b );
  m3_mulv( o->lk[vp->la].R, o->lk[vp->la].v+3, wa ); m3_mulv( o->lk[vp->lb].R, o->lk[vp->lb].v+3, wb );
  v3_sub( a, b, v ); v3_sub( wa, wb, v+3 );
}
static void vol_rel_acc6(const rkfdOracle *o, const VolPair *vp, double *r)
{
  double a[3], b[3], wa[3], wb[3];
  link_point_acc( &o->lk[vp->la], vp->center, a ); link_point_acc( &o->lk[vp->lb], vp->center, b );
  m3_mulv( o->lk[vp->la].R, o->lk[vp->la].a+3, wa ); m3_mulv( o->lk[vp->lb].R, o->lk[vp->lb].a+3, wb );
  v3_sub( a, b, r ); v3_sub( wa, wb, r+3 );
}
/* a world torque on link i */
static void ext_add_torque(rkfdOracle *o, int i, const double *tw, double sign)
{
  double t[3]; int k;
  m3_tmulv( o->lk[i].R, tw, t );
  for( k=0; k<3; k++ ) o->ext[6*i+3+k] += sign*t[k];
}

/* _rkFDSolverRelationAccForce (reference src/rkfd_volume.c:176-211) with _rkFDSolverBiasAcc (:143-154) and
 * _rkFDSolverRelativeAcc (:156-174): b = free 6-D relative accelerations at the pair centres; A column by column, the
 * response to a unit world force (i < 3) / torque (i >= 3) at the centre, + on cell[0], - on cell[1] */
static void vol_relation_acc_force(rkfdOracle *o, double *A, double *b, double *t)
{
  const rkfdModel *m = o->m;
  const int np = o->nvp, n = 6*np;
  int c, i, r, k;
  aba_backward_full( o );
  aba_forward( o, o->acc );
  aba_save_bias( o );
  for( c=0; c<np; c++ ) vol_rel_acc6( o, &o->vp[c], &b[6*c] );
  for( c=0; c<np; c++ ){
    const VolPair *vp = &o->vp[c];
    for( i=0; i<6; i++ ){
      double e[3] = {0,0,0};
      e[i%3] = 1.0;
      if( i < 3 ){
        ext_add( o, vp->la, vp->center, e,  1.0 );
        ext_add( o, vp->lb, vp->center, e, -1.0 );
      } else {
        ext_add_torque( o, vp->la, e,  1.0 );
        ext_add_torque( o, vp->lb, e, -1.0 );
      }
      aba_bias_path( o, vp->la );
      aba_bias_path( o, vp->lb );
      aba_forward( o, o->acc );
      for( r=0; r<np; r++ ){
        const VolPair *vr = &o->vp[r];
        int chA = m->chain[vr->la], chB = m->chain[vr->lb], pA = m->chain[vp->la], pB = m->chain[vp->lb];
        if( chA != pA && chA != pB && chB != pA && chB != pB ){
          for( k=0; k<6; k++ ) t[6*r+k] = 0;
        } else {
          double av[6];
          vol_rel_acc6( o, vr, av );
          for( k=0; k<6; k++ ) t[6*r+k] = av[k] - b[6*r+k];
        }
      }
      for( r=0; r<n; r++ ) A[n*r+6*c+i] = t[r];
      aba_restore_bias( o );
    }
  }
}

/* ------------------------------------------------------------------------ */
/* constraint on each contact volume (reference src/rkfd_volume.c:232-491) */
static double vol_tri_area(double (*p)[3])
{
  double e1[3], e2[3], x[3];
  v3_sub( p[1], p[0], e1 ); v3_sub( p[2], p[0], e2 ); v3_cross( e1, e2, x );
  return 0.5*v3_norm( x );
}
static void vol_mid_points(double (*p)[3], double (*pm)[3])
{
  int k;
  for( k=0; k<3; k++ ){ pm[0][k] = 0.5*( p[0][k]+p[1][k] ); pm[1][k] = 0.5*( p[1][k]+p[2][k] ); pm[2][k] = 0.5*( p[2][k]+p[0][k] ); }
}
/* _rkFDSolverConstraintAddQ (:279-294): q (6x6 row-major, (lin,ang) blocks) += area integral of [ 1  -[p x] ; [p x]  -[p x][p x] ] */
static void vol_add_q(double (*p)[3], double (*pm)[3], double s, double *q)
{
  double pc[3], k = s/3.0, mm[9] = {0,0,0,0,0,0,0,0,0};
  int i, a, b;
  for( a=0; a<3; a++ ) pc[a] = k*( p[0][a] + p[1][a] + p[2][a] );
  {
    const double px[9] = { 0,-pc[2],pc[1], pc[2],0,-pc[0], -pc[1],pc[0],0 };
    for( a=0; a<3; a++ ){
      q[6*a+a] += s;
      for( b=0; b<3; b++ ){ q[6*( 3+a )+b] += px[3*a+b]; q[6*a+3+b] -= px[3*a+b]; }
    }
  }
  for( i=0; i<3; i++ ){
    const double *v = pm[i];
    const double vx[9] = { 0,-v[2],v[1], v[2],0,-v[0], -v[1],v[0],0 };
    double t[9];
    m3_mul( vx, vx, t );
    for( a=0; a<9; a++ ) mm[a] += t[a];
  }
  for( a=0; a<3; a++ ) for( b=0; b<3; b++ ) q[6*( 3+a )+3+b] -= k*mm[3*a+b];
}
/* _rkFDSolverConstraintDepth (:296-310) with ...MidDepth (:232-241); scales pm in place as the reference does */
static void vol_depth(double (*pm)[3], const double *h, double K, double s, const double *norm, double *cc)
{
  const double k = K*s/6.0;
  const double hm[3] = { k*( h[0]+h[1] ), k*( h[1]+h[2] ), k*( h[0]+h[2] ) }, hc = k*( h[0]+h[1]+h[2] )*2;
  int i;
  v3_mul( norm, -hc, cc ); v3_zero( cc+3 );
  for( i=0; i<3; i++ ){
    double t[3];
    v3_mul( pm[i], hm[i], pm[i] );
    v3_cross( norm, pm[i], t );
    v3_add( cc+3, t, cc+3 );
  }
}
/* _rkFDSolverConstraintInnerPoint (:331-348) */
static void vol_inner_point(const double *p1, const double *p2, double h1, double h2, double *pp)
{
  int k;
  if( is_tiny( h1 ) ){ v3_copy( p1, pp ); return; }
  if( is_tiny( h2 ) ){ v3_copy( p2, pp ); return; }
  if( is_tiny( h2 - h1 ) ){ for( k=0; k<3; k++ ) pp[k] = 0.5*( p1[k]+p2[k] ); return; }
  v3_mul( p1, h2/( h2 - h1 ), pp );
  v3_cat( pp, h1/( h1 - h2 ), p2 );
}
/* _rkFDSolverSetContactPlane (:350-374) */
static void vol_set_contact_plane(VolPair *vp, const double *p, const double *fnorm)
{
  double tmpv[3], nn[3], l; int k;
  v3_copy( fnorm, tmpv ); v3_cat( tmpv, -v3_dot( fnorm, vp->norm ), vp->norm );
  if( fabs( tmpv[0] ) < TOL && fabs( tmpv[1] ) < TOL && fabs( tmpv[2] ) < TOL ) return;      /* zVec3DIsTiny */
  l = v3_norm( tmpv );
  v3_mul( tmpv, -1.0/l, nn );
  for( k=0; k<vp->ncp; k++ ){
    VolCP *c2 = &vp->cp[k];
    double d[3], x[3];
    v3_sub( nn, c2->n, d );
    if( !( fabs( d[0] ) < 1e-8 && fabs( d[1] ) < 1e-8 && fabs( d[2] ) < 1e-8 ) ) continue;
    v3_sub( c2->v, p, d );
    if( !( fabs( v3_dot( nn, d ) ) < 1e-8 ) ) continue;
    v3_cross( nn, d, x );
    if( v3_dot( vp->norm, x ) > 0.0 ) v3_copy( p, c2->v );
    return;
  }
  if( vp->ncp == VOL_MAXCP ) return;
  v3_copy( p, vp->cp[vp->ncp].v ); v3_copy( nn, vp->cp[vp->ncp].n );
  vp->ncp++;
}
/* __rk_fd_plane_cmp (:376-395): the angle of a condition's normal from axis[1], clockwise about the contact normal */
static double vol_plane_angle(const VolPair *vp, const double *pn)
{
  double tmp[3];
  v3_cross( vp->axis+3, pn, tmp );
  return v3_dot( tmp, vp->axis ) > 0 ? atan2( -v3_norm( tmp ), v3_dot( vp->axis+3, pn ) ) : atan2( v3_norm( tmp ), v3_dot( vp->axis+3, pn ) );
}

/* _rkFDSolverConstraint (:397-491) */
static void vol_constraint(rkfdOracle *o, VolPair *vp, double *q, double *c)
{
  const double K = o->m->ci_k[vp->ci];
  int i, j, k;
  memset( q, 0, sizeof(double)*36 ); memset( c, 0, sizeof(double)*6 );
  vp->ncp = 0;
  for( i=0; i<vp->ntri; i++ ){
    const double *face = &vp->tri[12*i], *fnorm = face+9;
    double h[3], s, pf[3][3], p[3][3], pm[3][3], pp[2][3], cc[6];
    int stp[3] = {0,0,0}, st = 0, neg = 0;
    for( j=0; j<3; j++ ){
      v3_sub( face+3*j, vp->center, pf[j] );
      h[j] = v3_dot( vp->norm, pf[j] );
      v3_copy( pf[j], p[j] ); v3_cat( p[j], -h[j], vp->norm );
    }
    vol_mid_points( p, pm );
    s = vol_tri_area( p );
    vol_add_q( p, pm, s, q );
    vol_depth( pm, h, K, s, vp->norm, cc );
    /* _rkFDSolverConstraintSignDepth (:312-329) */
    for( j=0; j<3; j++ ){
      if( h[j] > TOL ){ st += 1 << ( j*2 ); stp[1] = j; }
      else if( h[j] < -TOL ){ st += 1 << ( j*2+1 ); stp[2] = j; }
      else stp[0] = j;
    }
    switch( st ){
    case 0x01: case 0x04: case 0x10: case 0x05: case 0x11: case 0x14:
      vol_set_contact_plane( vp, pf[stp[0]], fnorm );
      /* fall through */
    case 0x15:
      for( k=0; k<6; k++ ) c[k] += cc[k];
      continue;
    case 0x02: case 0x08: case 0x20: case 0x0a: case 0x22: case 0x28:
      vol_set_contact_plane( vp, pf[stp[0]], fnorm );
      /* fall through */
    case 0x2a:
      for( k=0; k<6; k++ ) c[k] -= cc[k];
      continue;
    case 0x24: case 0x12: case 0x09:
      for( k=0; k<6; k++ ) c[k] += cc[k];
      vol_inner_point( pf[stp[1]], pf[stp[2]], h[stp[1]], h[stp[2]], p[stp[1]] );
      h[stp[1]] = 0.0;
      v3_copy( p[stp[0]], pp[0] ); v3_copy( p[stp[1]], pp[1] );
      break;
    case 0x06: case 0x21: case 0x18:
      for( k=0; k<6; k++ ) c[k] += cc[k];
      vol_inner_point( pf[stp[1]], pf[stp[2]], h[stp[1]], h[stp[2]], p[stp[2]] );
      h[stp[2]] = 0.0;
      v3_copy( p[stp[2]], pp[0] ); v3_copy( p[stp[0]], pp[1] );
      break;
    case 0x16: case 0x19: case 0x25:
      stp[0] = ( stp[2]+1 ) % 3; stp[1] = ( stp[0]+1 ) % 3;
      for( k=0; k<6; k++ ) c[k] += cc[k];
      vol_inner_point( pf[stp[2]], pf[stp[0]], h[stp[2]], h[stp[0]], p[stp[0]] );
      vol_inner_point( pf[stp[2]], pf[stp[1]], h[stp[2]], h[stp[1]], p[stp[1]] );
      h[stp[0]] = h[stp[1]] = 0.0;
      v3_copy( p[stp[0]], pp[0] ); v3_copy( p[stp[1]], pp[1] );
      break;
    case 0x1a: case 0x26: case 0x29:
      stp[0] = ( stp[1]+1 ) % 3; stp[2] = ( stp[0]+1 ) % 3;
      for( k=0; k<6; k++ ) c[k] -= cc[k];
      vol_inner_point( pf[stp[1]], pf[stp[0]], h[stp[1]], h[stp[0]], p[stp[0]] );
      vol_inner_point( pf[stp[1]], pf[stp[2]], h[stp[1]], h[stp[2]], p[stp[2]] );
      h[stp[0]] = h[stp[2]] = 0.0;
      v3_copy( p[stp[2]], pp[0] ); v3_copy( p[stp[0]], pp[1] );
      neg = 1;
      break;
    default:
      continue;
    }
    s = vol_tri_area( p );
    vol_mid_points( p, pm );
    vol_depth( pm, h, K, s, vp->norm, cc );
    for( k=0; k<6; k++ ) c[k] += ( neg ? 2.0 : -2.0 )*cc[k];
    vol_set_contact_plane( vp, pp[0], fnorm );
  }
  /* rkCDPlaneListQuickSort( &cpd->cplane, __rk_fd_plane_cmp, cpd->axis ) (:490) */
  for( i=0; i<vp->ncp; i++ ) vp->cp[i].th = vol_plane_angle( vp, vp->cp[i].n );
  for( i=1; i<vp->ncp; i++ ){
    VolCP x = vp->cp[i];
    for( j=i-1; j>=0 && !is_tiny( vp->cp[j].th - x.th ) && vp->cp[j].th > x.th; j-- ) vp->cp[j+1] = vp->cp[j];
    vp->cp[j+1] = x;
  }
}

/* ------------------------------------------------------------------------ */
/* zLPSolveSimplex [UNVERIFIED-DEP]: min c'x s.t. Ax = b (m rows), x >= 0 (n columns): two-phase tableau simplex.  c == NULL: phase 1 only (zLPFeasibleBase).  Returns 1 when an optimal (feasible) vertex was found. */
#define LP_EPS 1e-10
static int vol_lp(int mr, int n, const double *A, const double *b, const double *c, double *x)
{
  const int nt = n + mr, ld = nt + 1;
  double *T = (double *)malloc( sizeof(double)*( mr+1 )*ld ), *cost = (double *)malloc( sizeof(double)*ld );
  int *bas = (int *)malloc( sizeof(int)*mr ), i, j, k, ph, ok = 1, it;
  double scale = 0;
  for( i=0; i<mr; i++ ){
    const double sg = b[i] < 0 ? -1.0 : 1.0;
    for( j=0; j<n; j++ ) T[ld*i+j] = sg*A[n*i+j];
    for( j=0; j<mr; j++ ) T[ld*i+n+j] = i == j ? 1.0 : 0.0;
    T[ld*i+nt] = sg*b[i];
    bas[i] = n+i;
    if( fabs( b[i] ) > scale ) scale = fabs( b[i] );
  }
  for( ph=1; ph<=2 && ok; ph++ ){
    const int ncol = ph == 1 ? nt : n;
    if( ph == 2 && !c ) break;
    /* reduced costs of this phase */
    for( j=0; j<=nt; j++ ){
      double r = ph == 1 ? ( j >= n && j < nt ? 1.0 : 0.0 ) : ( j < n ? c[j] : 0.0 );
      for( i=0; i<mr; i++ ){
        const double cb = ph == 1 ? ( bas[i] >= n ? 1.0 : 0.0 ) : ( bas[i] < n ? c[bas[i]] : 0.0 );
        r -= cb*T[ld*i+j];
      }
      cost[j] = r;
    }
    for( it=0; it<10000; it++ ){
      int col = -1, row = -1; double best = 0;
      /* entering column: the most negative reduced cost (lowest index among equals); after 64 pivots of a phase Bland's
       * rule (the first improving column), which cannot cycle */
      if( it < 64 ){
        double cm = -LP_EPS;
        for( j=0; j<ncol; j++ ) if( cost[j] < cm ){ cm = cost[j]; col = j; }
      } else
        for( j=0; j<ncol; j++ ) if( cost[j] < -LP_EPS ){ col = j; break; }
      if( col < 0 ) break;
      for( i=0; i<mr; i++ )
        if( T[ld*i+col] > LP_EPS ){
          const double r = T[ld*i+nt]/T[ld*i+col];
          if( row < 0 || r < best - 1e-15 || ( !( r > best + 1e-15 ) && bas[i] < bas[row] ) ){ row = i; best = r; }
        }
      if( row < 0 ){ ok = 0; break; }                                           /* unbounded */
      {
        const double pv = 1.0/T[ld*row+col];
        for( j=0; j<=nt; j++ ) T[ld*row+j] *= pv;
        for( i=0; i<mr; i++ ){
          const double fct = T[ld*i+col];
          if( i == row || fct == 0.0 ) continue;
          for( j=0; j<=nt; j++ ) T[ld*i+j] -= fct*T[ld*row+j];
        }
        { const double fct = cost[col]; for( j=0; j<=nt; j++ ) cost[j] -= fct*T[ld*row+j]; }
        bas[row] = col;
      }
    }
    if( it == 10000 ) ok = 0;
    if( ph == 1 && ok ){
      double art = 0;
      for( i=0; i<mr; i++ ) if( bas[i] >= n ) art += T[ld*i+nt];
      if( art > 1e-9*( 1.0 + scale ) ) ok = 0;                                  /* infeasible */
      else
        for( i=0; i<mr; i++ )                                                   /* drive the artificials left at zero out of the base */
          if( bas[i] >= n ){
            for( j=0; j<n; j++ ) if( fabs( T[ld*i+j] ) > 1e-9 ) break;
            if( j < n ){
              const double pv = 1.0/T[ld*i+j];
              for( k=0; k<=nt; k++ ) T[ld*i+k] *= pv;
              for( k=0; k<mr; k++ ){
                const double fct = T[ld*k+j];
                int jj;
                if( k == i || fct == 0.0 ) continue;
                for( jj=0; jj<=nt; jj++ ) T[ld*k+jj] -= fct*T[ld*i+jj];
              }
              bas[i] = j;
            }
          }
    }
  }
  if( ok ){
    for( j=0; j<n; j++ ) x[j] = 0;
    for( i=0; i<mr; i++ ) if( bas[i] < n ) x[bas[i]] = T[ld*i+nt];
  }
  free( T ); free( cost ); free( bas );
  return ok;
}

/* ------------------------------------------------------------------------ */
/* _rkFDSolverModifyNormForceCenterTrq (:573-578) */
static void vol_center_trq(VolPair *vp, const double *r, double fn)
{
  double *tq = vp->wrench+3, nt = v3_dot( vp->norm, tq );
  v3_mul( vp->norm, nt, tq );
  v3_cat( tq,  fn*v3_dot( vp->axis+6, r ), vp->axis+3 );
  v3_cat( tq, -fn*v3_dot( vp->axis+3, r ), vp->axis+6 );
}
/* _rkFDSolverModifyNormalForceCenter (:580-631): a centre of normal force outside the contact polygon is moved onto its
 * boundary (a margin of zTOL inside).  The window of four consecutive conditions runs cyclically over the sorted list. */
static void vol_modify_normal_force_center(rkfdOracle *o)
{
  int c, k;
  for( c=0; c<o->nvp; c++ ){
    VolPair *vp = &o->vp[c];
    const int n = vp->ncp;
    double fn = v3_dot( vp->norm, vp->wrench ), r0[3], r[3], dir[3], tmp[3], d, s;
    int flag = 0;
    if( fn < TOL ) continue;
    if( n < 1 ) continue;
    v3_mul( vp->axis+3, -v3_dot( vp->axis+6, vp->wrench+3 )/fn, r0 );
    v3_cat( r0, v3_dot( vp->axis+3, vp->wrench+3 )/fn, vp->axis+6 );
    for( k=0; k<n; k++ ){
      const VolCP *c0 = &vp->cp[( k+3*n-3 ) % n], *c1 = &vp->cp[( k+3*n-2 ) % n], *c2 = &vp->cp[( k+3*n-1 ) % n], *c3 = &vp->cp[k];
      v3_sub( c2->v, c1->v, dir );
      d = v3_dot( dir, dir );
      if( is_tiny( d ) ) continue;
      v3_sub( r0, c1->v, tmp );
      if( v3_dot( tmp, c1->n ) > TOL ) continue;
      s = v3_dot( dir, tmp )/d;
      if( s < TOL ){
        if( flag ) break;
        v3_sub( c0->v, c1->v, tmp ); v3_add( tmp, dir, tmp );
        v3_copy( c1->v, r ); v3_cat( r, TOL/v3_norm( tmp ), tmp );
        vol_center_trq( vp, r, fn );
        break;
      } else if( s < 1.0-TOL ){
        v3_copy( c1->v, r ); v3_cat( r, s, dir ); v3_cat( r, TOL, c1->n );
        vol_center_trq( vp, r, fn );
        break;
      } else {
        v3_sub( c3->v, c2->v, tmp ); v3_sub( tmp, dir, tmp );
        v3_copy( c2->v, r ); v3_cat( r, TOL/v3_norm( tmp ), tmp );
        vol_center_trq( vp, r, fn );
        flag = 1;
      }
    }
  }
}

/* relative velocity of cell[0] against cell[1] at world point p, tangential to the contact normal
 * (rkFDChainPointRelativeVel, reference src/rkfd_util.c:42-60, with the slide mode of either cell) */
static void vol_tangent_vel(const rkfdOracle *o, const VolPair *vp, const double *p, double *v)
{
  double a[3], b[3];
  link_point_vel( &o->lk[vp->la], p, a ); add_slide_vel( o, vp->sa, p, vp->norm, a );
  link_point_vel( &o->lk[vp->lb], p, b ); add_slide_vel( o, vp->sb, p, vp->norm, b );
  v3_sub( a, b, v );
  v3_cat( v, -v3_dot( vp->norm, v ), vp->norm );
}
/* _rkFDSolverPlaneVertSlideDir (:759-776) */
static void vol_slide_dir(const rkfdOracle *o, VolPair *vp, VolCP *cp)
{
  const rkfdModel *m = o->m;
  double p[3], v[3], nv;
  v3_add( vp->center, cp->v, p );
  vol_tangent_vel( o, vp, p, v );
  nv = v3_norm( v );
  if( is_tiny( nv ) ){ cp->s[0] = cp->s[1] = 0; return; }
  {
    const double w = kf_weight( m->friction_weight, nv )*m->ci_kf[vp->ci]/nv;
    cp->s[0] = -w*v3_dot( v, vp->axis+3 ); cp->s[1] = -w*v3_dot( v, vp->axis+6 );
  }
}
/* _rkFDSolverModifyWrenchKinetic (:830-843): the normal force is spread over the vertices of the contact polygon so that it
 * keeps its resultant and centre; every vertex slides with its own direction.  w = wrench in the pair frame
 * ( f.axis[0..2], n.axis[0..2] ); w[1], w[2], w[3] are replaced. */
static void vol_kinetic(rkfdOracle *o, VolPair *vp, double *w)
{
  const int n = vp->ncp;
  double *ma = (double *)malloc( sizeof(double)*3*n ), *mc = (double *)calloc( (size_t)n, sizeof(double) ), *mf = (double *)calloc( (size_t)n, sizeof(double) );
  double mb[3] = { w[0], w[4], w[5] }, wn[3];
  int k, i;
  for( k=0; k<n; k++ ){ ma[k] = 1.0; ma[n+k] = vp->cp[k].r[1]; ma[2*n+k] = -vp->cp[k].r[0]; }
  /* _rkFDSolverModifyWrenchKineticEvalFunc (:778-793) */
  for( i=0; i<3; i++ ) wn[i] = is_tiny( w[i+1] ) ? 0.0 : 1.0/w[i+1];
  for( k=0; k<n; k++ ){
    VolCP *cp = &vp->cp[k];
    vol_slide_dir( o, vp, cp );
    mc[k] = -wn[0]*cp->s[0] - wn[1]*cp->s[1] - wn[2]*( cp->r[0]*cp->s[1] - cp->r[1]*cp->s[0] );
  }
  if( !vol_lp( 3, n, ma, mb, mc, mf ) ){
    /* _rkFDSolverModifyWrenchKineticEvalFuncSafety (:795-812): only the resultant is kept (one equality row) */
    for( i=0; i<2; i++ ) wn[i] = is_tiny( w[i+3] ) ? 0.0 : 1.0/w[i+3];
    for( k=0; k<n; k++ ){
      VolCP *cp = &vp->cp[k];
      vol_slide_dir( o, vp, cp );
      mc[k] += wn[0]*cp->r[0] - wn[1]*cp->r[1];
    }
    if( !vol_lp( 1, n, ma, mb, mc, mf ) ) o->vol_lp_fail++;
  }
  /* _rkFDSolverModifyWrenchKineticTotalWrench (:814-828) */
  w[1] = w[2] = w[3] = 0;
  for( k=0; k<n; k++ ){
    const VolCP *cp = &vp->cp[k];
    const double fx = cp->s[0]*mf[k], fy = cp->s[1]*mf[k];
    w[1] += fx; w[2] += fy; w[3] += cp->r[0]*fy - cp->r[1]*fx;
  }
  free( ma ); free( mc ); free( mf );
}
/* _rkFDSolverModifyWrenchStatic (:677-688): can the wrench be written as forces inside the friction pyramids at the
 * vertices of the contact polygon? */
static int vol_static(rkfdOracle *o, VolPair *vp, const double *w)
{
  const rkfdModel *m = o->m;
  const int P = m->pyramid > 0 ? m->pyramid : 8, n = P*vp->ncp;
  const double mu = m->ci_sf[vp->ci], dth = 2.0*M_PI/P;
  double *ma = (double *)malloc( sizeof(double)*6*n ), *mf = (double *)malloc( sizeof(double)*n );
  double mb[6] = { w[0], w[4], w[5], w[1], w[2], w[3] }, th;
  int k, i, ret;
  for( k=0; k<vp->ncp; k++ )
    for( i=0, th=0.0; i<P; i++, th+=dth ){
      const int j = P*k+i;
      ma[j] = 1.0; ma[n+j] = vp->cp[k].r[1]; ma[2*n+j] = -vp->cp[k].r[0];
      ma[3*n+j] = mu*cos( th + 0.0 ); ma[4*n+j] = mu*sin( th + 0.0 );
      ma[5*n+j] = -( ma[2*n+j]*ma[4*n+j] + ma[n+j]*ma[3*n+j] );
    }
  ret = vol_lp( 6, n, ma, mb, NULL, mf );
  free( ma ); free( mf );
  return ret;
}
/* _rkFDSolverModifyWrenchSetForce (:858-867) */
static void vol_set_force(VolPair *vp, const double *w)
{
  int i;
  for( i=0; i<6; i++ ) vp->wrench[i] = 0;
  for( i=0; i<3; i++ ){ v3_cat( vp->wrench, w[i], vp->axis+3*i ); v3_cat( vp->wrench+3, w[i+3], vp->axis+3*i ); }
}
/* _rkFDSolverModifyWrench (:869-916) */
static void vol_modify_wrench(rkfdOracle *o, int doUpRef)
{
  const rkfdModel *m = o->m;
  int c, i, k;
  for( c=0; c<o->nvp; c++ ){
    VolPair *vp = &o->vp[c];
    double w[6], fn, fs, tl = 0;
    const double sf = m->ci_sf[vp->ci];
    if( vp->ncp == 0 ) continue;
    if( is_tiny( v3_dot( vp->wrench, vp->axis ) ) ) continue;
    for( i=0; i<3; i++ ){ w[i] = v3_dot( vp->wrench, vp->axis+3*i ); w[i+3] = v3_dot( vp->wrench+3, vp->axis+3*i ); }
    fn = w[0];
    fs = sqrt( w[1]*w[1] + w[2]*w[2] );
    /* _rkFDSolverPlaneVertPos (:700-713) */
    for( k=0; k<vp->ncp; k++ ){
      VolCP *cp = &vp->cp[k];
      double rl;
      cp->r[0] = v3_dot( cp->v, vp->axis+3 ); cp->r[1] = v3_dot( cp->v, vp->axis+6 );
      rl = sqrt( cp->r[0]*cp->r[0] + cp->r[1]*cp->r[1] );
      if( tl < rl ) tl = rl;
    }
    if( is_tiny( tl ) ){
      w[3] = w[4] = w[5] = 0;
      if( !is_tiny( fs ) && fs > sf*fn ){
        /* _rkFDSolverModifyWrenchKineticCenter (:715-731) */
        double v[3], nv;
        vol_tangent_vel( o, vp, vp->center, v );
        nv = v3_norm( v );
        if( is_tiny( nv ) ){ w[1] = w[2] = 0; }
        else {
          const double t = kf_weight( m->friction_weight, nv )*m->ci_kf[vp->ci]*w[0]/nv;
          w[1] = -t*v3_dot( v, vp->axis+3 ); w[2] = -t*v3_dot( v, vp->axis+6 );
        }
        if( doUpRef ) o->vp_type[vp->pair] = RKFD_KF;
      } else if( doUpRef ) o->vp_type[vp->pair] = RKFD_SF;
      vol_set_force( vp, w );
      continue;
    }
    if( ( !is_tiny( fs ) && fs > sf*fn ) || fabs( w[3] ) > tl*w[0] ){
      vol_kinetic( o, vp, w );
      if( doUpRef ) o->vp_type[vp->pair] = RKFD_KF;
      vol_set_force( vp, w );
    } else if( vol_static( o, vp, w ) ){
      if( doUpRef ) o->vp_type[vp->pair] = RKFD_SF;
    } else {
      vol_kinetic( o, vp, w );
      if( doUpRef ) o->vp_type[vp->pair] = RKFD_KF;
      vol_set_force( vp, w );
    }
  }
}

/* ------------------------------------------------------------------------ */
/* _rkFDSolverVolume (:939-957) */
static int volume_rigid(rkfdOracle *o, int doUpRef)
{
  const rkfdModel *m = o->m;
  const int np = o->nvp, n = 6*np;
  const double dt = m->dt;
  double *A = (double *)malloc( sizeof(double)*n*n ), *b = (double *)malloc( sizeof(double)*n ), *t = (double *)malloc( sizeof(double)*n );
  double *q = (double *)calloc( (size_t)n*n, sizeof(double) ), *cv = (double *)calloc( n, sizeof(double) ), *f = (double *)malloc( sizeof(double)*n );
  double *nf, *d, *init = (double *)calloc( n, sizeof(double) );
  int *idx, c, i, j, k, r, cnum, colnum = 0, io, jo, off;

  vol_relation_acc_force( o, A, b, t );
  /* _rkFDSolverBiasVel (:214-226) */
  for( r=0; r<n; r++ ) b[r] *= dt;
  for( c=0; c<np; c++ ){
    double vr[6];
    vol_rel_vel6( o, &o->vp[c], vr );
    for( k=0; k<6; k++ ) b[6*c+k] += vr[k];
  }
  /* _rkFDSolverQPCreate (:496-528): q = sum_p A_p' qv A_p + L, c = sum_p A_p' ( qv b_p + cv ) */
  for( c=0; c<np; c++ ){
    VolPair *vp = &o->vp[c];
    double tmpv[6];
    vol_constraint( o, vp, vp->q, vp->c );
    for( i=0; i<6; i++ )
      for( j=0; j<6; j++ ){
        const double e = vp->q[6*i+j];
        for( r=0; r<n; r++ ) for( k=0; k<n; k++ ) q[n*r+k] += e*A[n*( 6*c+i )+r]*A[n*( 6*c+j )+k];
      }
    for( i=0; i<6; i++ ){
      double s = 0;
      for( j=0; j<6; j++ ) s += vp->q[6*i+j]*b[6*c+j];
      tmpv[i] = vp->c[i] + s;
    }
    for( i=0; i<6; i++ ) for( r=0; r<n; r++ ) cv[r] += tmpv[i]*A[n*( 6*c+i )+r];
  }
  for( c=0; c<np; c++ ) for( i=0; i<6; i++ ) q[n*( 6*c+i )+6*c+i] += m->ci_l[o->vp[c].ci];
  /* _rkFDSolverCountContacts (:21-28), _rkFDSolverFrictionConstraint (:121-138) */
  for( c=0; c<np; c++ ) colnum += o->vp[c].ncp;
  cnum = np + colnum;
  if( cnum < 1 ) cnum = 1;      /* (np >= 1 here; said for the compiler's range analysis) */
  nf = (double *)calloc( (size_t)cnum*n, sizeof(double) ); d = (double *)calloc( (size_t)cnum, sizeof(double) ); idx = (int *)malloc( sizeof(int)*(size_t)cnum );
  io = 0; jo = 0;
  for( c=0; c<np; c++ ){
    const VolPair *vp = &o->vp[c];
    v3_copy( vp->norm, &nf[n*io+jo] ); io++;
    for( k=0; k<vp->ncp; k++ ){
      const VolCP *cp = &vp->cp[k];
      double *row = &nf[n*io+jo];
      v3_mul( vp->norm, -v3_dot( cp->n, cp->v ), row );
      v3_mul( vp->axis+3, v3_dot( cp->n, vp->axis+6 ), row+3 );
      v3_cat( row+3, -v3_dot( cp->n, vp->axis+3 ), vp->axis+6 );
      io++;
    }
    jo += 6;
  }
  /* _rkFDSolverQP (:544-548) from the start point of _rkFDSolverQPInit (:530-542): a unit normal force per pair */
  for( c=0; c<np; c++ ) v3_copy( o->vp[c].norm, &init[6*c] );
  o->last_qp_iter = qp_asm_ex( n, cnum, 0, q, cv, nf, d, init, f, idx );
  if( o->last_qp_iter < 0 ){ o->last_qp_iter = -o->last_qp_iter; o->qp_cycle_stops++; }
  for( r=0; r<n; r++ ) f[r] /= dt;
  /* _rkFDSolverSetForce (:552-568); the offset does not advance past a pair without contact-plane conditions, as in the reference */
  off = 0;
  for( c=0; c<np; c++ ){
    VolPair *vp = &o->vp[c];
    if( vp->ncp == 0 ){ for( k=0; k<6; k++ ) vp->wrench[k] = 0; continue; }
    for( k=0; k<6; k++ ) vp->wrench[k] = f[off+k];
    if( ( fabs( vp->wrench[0] ) < TOL && fabs( vp->wrench[1] ) < TOL && fabs( vp->wrench[2] ) < TOL ) || v3_dot( vp->wrench, vp->norm ) < TOL )
      for( k=0; k<6; k++ ) vp->wrench[k] = 0;
    off += 6;
  }
  vol_modify_normal_force_center( o );
  vol_modify_wrench( o, doUpRef );
  /* _rkFDSolverPushWrench (:919-936) */
  for( c=0; c<np; c++ ){
    const VolPair *vp = &o->vp[c];
    ext_add( o, vp->la, vp->center, vp->wrench,  1.0 ); ext_add_torque( o, vp->la, vp->wrench+3,  1.0 );
    ext_add( o, vp->lb, vp->center, vp->wrench, -1.0 ); ext_add_torque( o, vp->lb, vp->wrench+3, -1.0 );
  }
  free( A ); free( b ); free( t ); free( q ); free( cv ); free( f ); free( nf ); free( d ); free( init ); free( idx );
  return 0;
}

#endif /* RKFD_ORACLE_VOLUME_H */
