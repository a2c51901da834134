/* rkfd_ztk.c - minimal ZTK reader (see rkfd_ztk.h for scope).
 * Independent implementation; replaces RoKi's rkChainReadZTK /
 * rkContactInfoArrayReadZTK at the call sites reference src/rkfd_sim.c:229,264.
 */
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <math.h>
#include "rkfd_model.h"
#include "rkfd_ztk.h"

/* ------------------------------------------------------------------------ */
/* tokenizer: a ZTK file is a sequence of [tag] sections, each a sequence of
 * "key : value value ..." fields; braces, parentheses, commas and semicolons
 * are separators; '%' starts a comment; a field's values may span lines. */
typedef struct {
  char tag[RKFD_NAME_MAX];
  char key[RKFD_NAME_MAX];
  int nval;
  char **val;
  int cap;
} Field;

typedef struct {
  int nfield;
  Field *field;
  int cap;
} Doc;

static void field_add_val(Field *f, const char *s, int len)
{
  if( f->nval == f->cap ){
    f->cap = f->cap ? 2*f->cap : 8;
    f->val = (char **)realloc( f->val, sizeof(char*)*f->cap );
  }
  f->val[f->nval] = (char *)malloc( len+1 );
  memcpy( f->val[f->nval], s, len );
  f->val[f->nval][len] = '\0';
  f->nval++;
}

static Field *doc_new_field(Doc *d, const char *tag, const char *key)
{
  Field *f;
  if( d->nfield == d->cap ){
    d->cap = d->cap ? 2*d->cap : 64;
    d->field = (Field *)realloc( d->field, sizeof(Field)*d->cap );
  }
  f = &d->field[d->nfield++];
  memset( f, 0, sizeof(Field) );
  strncpy( f->tag, tag, RKFD_NAME_MAX-1 );
  strncpy( f->key, key, RKFD_NAME_MAX-1 );
  return f;
}

static void doc_free(Doc *d)
{
  int i, j;
  for( i=0; i<d->nfield; i++ ){
    for( j=0; j<d->field[i].nval; j++ ) free( d->field[i].val[j] );
    free( d->field[i].val );
  }
  free( d->field );
}

static int is_sep(int c)
{
  return isspace(c) || c == ',' || c == '{' || c == '}' || c == '(' || c == ')' || c == ';';
}

static void tokenize_values(Field *f, const char *s)
{
  const char *p = s, *q;
  while( *p ){
    while( *p && is_sep((unsigned char)*p) ) p++;
    if( !*p ) break;
    q = p;
    while( *q && !is_sep((unsigned char)*q) ) q++;
    field_add_val( f, p, (int)(q-p) );
    p = q;
  }
}

static int doc_read(Doc *d, const char *filename)
{
  FILE *fp;
  char line[4096], tag[RKFD_NAME_MAX] = "", key[RKFD_NAME_MAX];
  char *p, *c, *k, *ke;
  Field *cur = NULL;

  memset( d, 0, sizeof(Doc) );
  if( !( fp = fopen( filename, "r" ) ) ){
    fprintf( stderr, "rkfd: cannot open file %s\n", filename );
    return -1;
  }
  while( fgets( line, sizeof(line), fp ) ){
    if( ( c = strchr( line, '%' ) ) ) *c = '\0';
    p = line;
    while( *p && isspace((unsigned char)*p) ) p++;
    if( !*p ) continue;
    if( *p == '[' ){
      c = strchr( p, ']' );
      if( !c ) continue;
      *c = '\0';
      strncpy( tag, p+1, RKFD_NAME_MAX-1 );
      tag[RKFD_NAME_MAX-1] = '\0';
      /* a tag marks a section boundary even when it has no fields */
      cur = doc_new_field( d, tag, "" );
      cur = NULL;
      continue;
    }
    /* "key :" prefix?  the key is one identifier-like token before the first ':' */
    c = strchr( p, ':' );
    k = p; ke = NULL;
    if( c ){
      ke = k;
      while( ke < c && !isspace((unsigned char)*ke) ) ke++;
      { char *t = ke; while( t < c && isspace((unsigned char)*t) ) t++; if( t != c ) ke = NULL; }
      if( ke == k ) ke = NULL;
    }
    if( ke ){
      int len = (int)(ke-k);
      if( len > RKFD_NAME_MAX-1 ) len = RKFD_NAME_MAX-1;
      memcpy( key, k, len ); key[len] = '\0';
      cur = doc_new_field( d, tag, key );
      tokenize_values( cur, c+1 );
    } else if( cur ){
      tokenize_values( cur, p );
    }
  }
  fclose( fp );
  return 0;
}

static double fval(const Field *f, int i)
{
  return i < f->nval ? strtod( f->val[i], NULL ) : 0.0;
}

static void sval(const Field *f, int i, char *dst)
{
  dst[0] = '\0';
  if( i < f->nval ){ strncpy( dst, f->val[i], RKFD_NAME_MAX-1 ); dst[RKFD_NAME_MAX-1] = '\0'; }
}

/* ------------------------------------------------------------------------ */
/* shapes */
static void v3_sub(const double *a, const double *b, double *c){ c[0]=a[0]-b[0]; c[1]=a[1]-b[1]; c[2]=a[2]-b[2]; }
static void v3_cross(const double *a, const double *b, double *c)
{
  double x = a[1]*b[2]-a[2]*b[1], y = a[2]*b[0]-a[0]*b[2], z = a[0]*b[1]-a[1]*b[0];
  c[0]=x; c[1]=y; c[2]=z;
}
static double v3_dot(const double *a, const double *b){ return a[0]*b[0]+a[1]*b[1]+a[2]*b[2]; }

/* build the deduplicated outward face planes of a convex polyhedron from its triangles; the triangles are kept
 * (writer, mass properties).  -1 when a face names a vertex that does not exist. */
static int shape_build_planes(rkfdShape *s, int nface, const int *face)
{
  int i, j, k;
  double cen[3] = {0,0,0}, e1[3], e2[3], n[4], len;

  for( i=0; i<3*nface; i++ )
    if( face[i] < 0 || face[i] >= s->nvert ) return -1;
  free( s->face );
  s->nface = nface;
  s->face = (int *)malloc( sizeof(int)*3*( nface > 0 ? nface : 1 ) );
  if( nface > 0 ) memcpy( s->face, face, sizeof(int)*3*nface );
  for( i=0; i<s->nvert; i++ )
    for( k=0; k<3; k++ ) cen[k] += s->vert[3*i+k] / s->nvert;
  s->plane = (double *)malloc( sizeof(double)*4*( nface > 0 ? nface : 1 ) );
  s->nplane = 0;
  for( i=0; i<nface; i++ ){
    const double *a = &s->vert[3*face[3*i]], *b = &s->vert[3*face[3*i+1]], *c = &s->vert[3*face[3*i+2]];
    v3_sub( b, a, e1 ); v3_sub( c, a, e2 );
    v3_cross( e1, e2, n );
    len = sqrt( v3_dot( n, n ) );
    if( len < 1e-14 ) continue;
    for( k=0; k<3; k++ ) n[k] /= len;
    n[3] = v3_dot( n, a );
    if( v3_dot( n, cen ) - n[3] > 0 ){ for( k=0; k<4; k++ ) n[k] = -n[k]; }
    for( j=0; j<s->nplane; j++ ){
      const double *q = &s->plane[4*j];
      if( fabs(q[0]-n[0]) < 1e-9 && fabs(q[1]-n[1]) < 1e-9 && fabs(q[2]-n[2]) < 1e-9 && fabs(q[3]-n[3]) < 1e-9 ) break;
    }
    if( j == s->nplane ){
      memcpy( &s->plane[4*s->nplane], n, sizeof(double)*4 );
      s->nplane++;
    }
  }
  s->convex = 1;
  for( i=0; i<s->nplane && s->convex; i++ )
    for( j=0; j<s->nvert; j++ )
      if( v3_dot( &s->plane[4*i], &s->vert[3*j] ) - s->plane[4*i+3] > 1e-9 ){ s->convex = 0; break; }
  return 0;
}

static void shape_make_box(rkfdShape *s, const double *center, double dx, double dy, double dz)
{
  static const double sg[8][3] = {
    { 1, 1, 1}, {-1, 1, 1}, {-1,-1, 1}, { 1,-1, 1},
    { 1, 1,-1}, {-1, 1,-1}, {-1,-1,-1}, { 1,-1,-1} };
  static const int tri[12*3] = {
    0,1,2, 0,2,3, 0,4,5, 0,5,1, 1,5,6, 1,6,2, 2,6,7, 2,7,3, 3,7,4, 3,4,0, 7,6,5, 7,5,4 };
  int i;
  s->nvert = 8;
  s->vert = (double *)malloc( sizeof(double)*24 );
  for( i=0; i<8; i++ ){
    s->vert[3*i  ] = center[0] + 0.5*dx*sg[i][0];
    s->vert[3*i+1] = center[1] + 0.5*dy*sg[i][1];
    s->vert[3*i+2] = center[2] + 0.5*dz*sg[i][2];
  }
  shape_build_planes( s, 12, tri );
}

/* an orthonormal pair (u, v) perpendicular to the unit vector a */
static void perp_pair(const double *a, double *u, double *v)
{
  int k = 0;
  double e[3] = {0,0,0}, d, l;
  if( fabs(a[1]) < fabs(a[k]) ) k = 1;
  if( fabs(a[2]) < fabs(a[k]) ) k = 2;
  e[k] = 1.0;
  d = v3_dot( e, a );
  u[0] = e[0]-d*a[0]; u[1] = e[1]-d*a[1]; u[2] = e[2]-d*a[2];
  l = sqrt( v3_dot( u, u ) );
  u[0] /= l; u[1] /= l; u[2] /= l;
  v3_cross( a, u, v );
}
/* the curved primitives as the convex polyhedra a tessellation with `div` divisions gives
 * (the role of Zeo's zShape3DToPH in rkCDChainReg; the vertex placement is this reader's own) */
static void shape_make_cylinder(rkfdShape *s, const double *c0, const double *c1, double r, int div, int cone)
{
  double a[3], u[3], v[3], l;
  int i, k, nf = 0, *tri;
  v3_sub( c1, c0, a );
  l = sqrt( v3_dot( a, a ) );
  if( l < 1e-14 ){ a[0] = 0; a[1] = 0; a[2] = 1; } else { a[0] /= l; a[1] /= l; a[2] /= l; }
  perp_pair( a, u, v );
  s->nvert = cone ? div+1 : 2*div;
  s->vert = (double *)malloc( sizeof(double)*3*s->nvert );
  for( i=0; i<div; i++ ){
    const double th = 2.0*3.14159265358979323846*i/div, cs = r*cos( th ), sn = r*sin( th );
    for( k=0; k<3; k++ ){
      s->vert[3*i+k] = c0[k] + cs*u[k] + sn*v[k];
      if( !cone ) s->vert[3*( div+i )+k] = c1[k] + cs*u[k] + sn*v[k];
    }
  }
  if( cone ) for( k=0; k<3; k++ ) s->vert[3*div+k] = c1[k];
  tri = (int *)malloc( sizeof(int)*3*( 4*div ) );
  for( i=0; i<div; i++ ){
    const int j = ( i+1 )%div;
    if( cone ){ tri[3*nf] = i; tri[3*nf+1] = j; tri[3*nf+2] = div; nf++; }
    else {
      tri[3*nf] = i; tri[3*nf+1] = j; tri[3*nf+2] = div+j; nf++;
      tri[3*nf] = i; tri[3*nf+1] = div+j; tri[3*nf+2] = div+i; nf++;
    }
  }
  for( i=1; i+1<div; i++ ){
    tri[3*nf] = 0; tri[3*nf+1] = i+1; tri[3*nf+2] = i; nf++;
    if( !cone ){ tri[3*nf] = div; tri[3*nf+1] = div+i; tri[3*nf+2] = div+i+1; nf++; }
  }
  shape_build_planes( s, nf, tri );
  free( tri );
}
static void shape_make_sphere(rkfdShape *s, const double *c, double r, int div)
{
  const int nr = div/2 - 1 > 1 ? div/2 - 1 : 1;     /* latitude rings between the poles */
  int i, j, nf = 0, *tri;
  s->nvert = nr*div + 2;
  s->vert = (double *)malloc( sizeof(double)*3*s->nvert );
  for( i=0; i<nr; i++ ){
    const double ph = 3.14159265358979323846*( i+1 )/( nr+1 );
    for( j=0; j<div; j++ ){
      const double th = 2.0*3.14159265358979323846*j/div;
      double *q = &s->vert[3*( i*div+j )];
      q[0] = c[0] + r*sin( ph )*cos( th ); q[1] = c[1] + r*sin( ph )*sin( th ); q[2] = c[2] + r*cos( ph );
    }
  }
  { double *n = &s->vert[3*nr*div], *so = n + 3; n[0] = c[0]; n[1] = c[1]; n[2] = c[2]+r; so[0] = c[0]; so[1] = c[1]; so[2] = c[2]-r; }
  tri = (int *)malloc( sizeof(int)*3*( 2*nr*div ) );
  for( j=0; j<div; j++ ){
    const int j1 = ( j+1 )%div;
    tri[3*nf] = nr*div; tri[3*nf+1] = j; tri[3*nf+2] = j1; nf++;
    tri[3*nf] = nr*div+1; tri[3*nf+1] = ( nr-1 )*div+j1; tri[3*nf+2] = ( nr-1 )*div+j; nf++;
    for( i=0; i+1<nr; i++ ){
      tri[3*nf] = i*div+j; tri[3*nf+1] = ( i+1 )*div+j; tri[3*nf+2] = ( i+1 )*div+j1; nf++;
      tri[3*nf] = i*div+j; tri[3*nf+1] = ( i+1 )*div+j1; tri[3*nf+2] = i*div+j1; nf++;
    }
  }
  shape_build_planes( s, nf, tri );
  free( tri );
}
/* a polyhedron given as a planar loop swept along a vector (Zeo's `loop:` + `prism:` keys, reference example/model/puma.ztk:63-88):
 * loop: <axis x|y|z> <offset> then points (two coordinates in the plane perpendicular to the axis) and
 * `arc cw|ccw <radius> <div>` between the point before and the point after.  Returns -1 on a malformed loop.
 * [UNVERIFIED-DEP: which side `cw` bulges to - taken as clockwise seen against the axis direction] */
static int shape_make_prism(rkfdShape *s, const Field *loop, const double *sweep)
{
  double (*pt)[2] = NULL, off;
  int np = 0, cap = 0, i, k, ax, nf = 0, *tri;
  int pend_arc = 0, arc_cw = 0, arc_div = 0; double arc_r = 0;
  if( loop->nval < 2 ) return -1;
  ax = loop->val[0][0] == 'x' ? 0 : ( loop->val[0][0] == 'y' ? 1 : 2 );
  off = strtod( loop->val[1], NULL );
#define PUSH_PT(x_, y_) do{ if( np == cap ){ cap = cap ? 2*cap : 64; pt = (double (*)[2])realloc( pt, sizeof(double)*2*cap ); } pt[np][0] = (x_); pt[np][1] = (y_); np++; }while(0)
  for( i=2; i<loop->nval; ){
    if( strcmp( loop->val[i], "arc" ) == 0 ){
      if( i+3 >= loop->nval || np == 0 ){ free( pt ); return -1; }
      arc_cw = strcmp( loop->val[i+1], "cw" ) == 0; arc_r = strtod( loop->val[i+2], NULL ); arc_div = atoi( loop->val[i+3] );
      pend_arc = 1; i += 4;
      continue;
    }
    if( i+1 >= loop->nval ){ free( pt ); return -1; }
    {
      const double x = strtod( loop->val[i], NULL ), y = strtod( loop->val[i+1], NULL );
      i += 2;
      if( pend_arc && arc_div > 1 ){
        const double px = pt[np-1][0], py = pt[np-1][1], dx = x-px, dy = y-py, h = 0.5*sqrt( dx*dx+dy*dy );
        if( h > 1e-14 && arc_r >= h ){
          const double d = sqrt( arc_r*arc_r - h*h ), ux = dx/( 2*h ), uy = dy/( 2*h );
          const double cx = 0.5*( px+x ) + ( arc_cw ? 1.0 : -1.0 )*d*uy, cy = 0.5*( py+y ) - ( arc_cw ? 1.0 : -1.0 )*d*ux;
          double a0 = atan2( py-cy, px-cx ), a1 = atan2( y-cy, x-cx );
          if( arc_cw ){ while( a1 > a0 ) a1 -= 2*3.14159265358979323846; } else { while( a1 < a0 ) a1 += 2*3.14159265358979323846; }
          for( k=1; k<arc_div; k++ ){
            const double a = a0 + ( a1-a0 )*k/arc_div;
            PUSH_PT( cx + arc_r*cos( a ), cy + arc_r*sin( a ) );
          }
        }
      }
      pend_arc = 0;
      PUSH_PT( x, y );
    }
  }
#undef PUSH_PT
  if( np < 3 ){ free( pt ); return -1; }
  s->nvert = 2*np;
  s->vert = (double *)malloc( sizeof(double)*3*s->nvert );
  for( i=0; i<np; i++ ){
    double q[3];
    q[ax] = off; q[( ax+1 )%3] = pt[i][0]; q[( ax+2 )%3] = pt[i][1];
    for( k=0; k<3; k++ ){ s->vert[3*i+k] = q[k]; s->vert[3*( np+i )+k] = q[k] + sweep[k]; }
  }
  free( pt );
  tri = (int *)malloc( sizeof(int)*3*( 4*np ) );
  for( i=0; i<np; i++ ){
    const int j = ( i+1 )%np;
    tri[3*nf] = i; tri[3*nf+1] = j; tri[3*nf+2] = np+j; nf++;
    tri[3*nf] = i; tri[3*nf+1] = np+j; tri[3*nf+2] = np+i; nf++;
  }
  for( i=1; i+1<np; i++ ){
    tri[3*nf] = 0; tri[3*nf+1] = i+1; tri[3*nf+2] = i; nf++;
    tri[3*nf] = np; tri[3*nf+1] = np+i; tri[3*nf+2] = np+i+1; nf++;
  }
  shape_build_planes( s, nf, tri );
  free( tri );
  return 0;
}
/* volume, first and second moments of a closed triangle mesh about the origin of its frame (signed tetrahedra, faces
 * oriented outwards by the centroid test - the shapes are convex) */
static void shape_moments(const rkfdShape *s, double *vol, double *first, double *second /* xx yy zz xy yz zx */)
{
  int i, k;
  double cen[3] = {0,0,0};
  *vol = 0; first[0] = first[1] = first[2] = 0;
  for( k=0; k<6; k++ ) second[k] = 0;
  for( i=0; i<s->nvert; i++ ) for( k=0; k<3; k++ ) cen[k] += s->vert[3*i+k]/s->nvert;
  for( i=0; i<s->nface; i++ ){
    const double *a = &s->vert[3*s->face[3*i]], *b = &s->vert[3*s->face[3*i+1]], *c = &s->vert[3*s->face[3*i+2]];
    double e1[3], e2[3], n[3], ac[3], v6, sg;
    v3_sub( b, a, e1 ); v3_sub( c, a, e2 ); v3_cross( e1, e2, n ); v3_sub( a, cen, ac );
    sg = v3_dot( n, ac ) >= 0 ? 1.0 : -1.0;
    v3_cross( b, c, e1 );
    v6 = sg*v3_dot( a, e1 );                      /* 6 x signed volume of the tetrahedron (0, a, b, c) */
    *vol += v6/6.0;
    for( k=0; k<3; k++ ) first[k] += v6*( a[k]+b[k]+c[k] )/24.0;
    {
      static const int ix[6][2] = { {0,0}, {1,1}, {2,2}, {0,1}, {1,2}, {2,0} };
      for( k=0; k<6; k++ ){
        const int p = ix[k][0], q = ix[k][1];
        second[k] += v6*( a[p]*a[q] + b[p]*b[q] + c[p]*c[q] + ( a[p]+b[p]+c[p] )*( a[q]+b[q]+c[q] ) )/120.0;
      }
    }
  }
}

/* ------------------------------------------------------------------------ */
static int find_name(const char *name, const char *base, size_t stride, int n)
{
  int i;
  for( i=0; i<n; i++ )
    if( strcmp( base + stride*i, name ) == 0 ) return i;
  return -1;
}

static int parse_jointtype(const char *s)
{
  if( strncmp( s, "revol", 5 ) == 0 ) return RKFD_JOINT_REVOL;
  if( strncmp( s, "prism", 5 ) == 0 ) return RKFD_JOINT_PRISM;
  if( strcmp( s, "float" ) == 0 ) return RKFD_JOINT_FLOAT;
  if( strncmp( s, "spher", 5 ) == 0 ) return RKFD_JOINT_SPHER;
  if( strncmp( s, "breakablefloat", 14 ) == 0 || strcmp( s, "brfloat" ) == 0 ) return RKFD_JOINT_BRFLOAT;
  if( strcmp( s, "fixed" ) == 0 || strcmp( s, "fix" ) == 0 ) return RKFD_JOINT_FIXED;
  return -1;
}

rkfdChainDesc *rkfdChainReadZTK(const char *filename)
{
  Doc d;
  rkfdChainDesc *c;
  int i, j, k, nl = 0, ns = 0, nm = 0;
  int *order, *newidx;
  rkfdLinkDesc *sorted;
  /* scratch for the current shape */
  int cur_shape = -1, shape_type = 0, nface = 0, facecap = 0, *face = NULL, vcap = 0, ncenter = 0, div = 0, bad = 0;
  double center[2][3] = {{0,0,0},{0,0,0}}, bx = 0, by = 0, bz = 0, radius = 0, apex[3] = {0,0,0}, sweep[3] = {0,0,0};
  const Field *loopf = NULL;
  int cur_link = -1, cur_motor = -1, in_init = 0;

  if( doc_read( &d, filename ) < 0 ) return NULL;
  c = (rkfdChainDesc *)calloc( 1, sizeof(rkfdChainDesc) );
  for( i=0; i<d.nfield; i++ ){
    if( d.field[i].key[0] ) continue;
    if( strcmp( d.field[i].tag, "roki::link" ) == 0 ) nl++;
    if( strcmp( d.field[i].tag, "zeo::shape" ) == 0 ) ns++;
    if( strcmp( d.field[i].tag, "roki::motor" ) == 0 ) nm++;
  }
  if( nl == 0 ){
    fprintf( stderr, "rkfd: no [roki::link] in %s\n", filename );
    doc_free( &d ); free( c );
    return NULL;
  }
  c->link  = (rkfdLinkDesc *)calloc( nl, sizeof(rkfdLinkDesc) );
  c->shape = (rkfdShape *)calloc( ns ? ns : 1, sizeof(rkfdShape) );
  c->motor = (rkfdMotor *)calloc( nm ? nm : 1, sizeof(rkfdMotor) );

#define FINISH_SHAPE() do{ \
    if( cur_shape >= 0 ){ \
      rkfdShape *s_ = &c->shape[cur_shape]; \
      const int dv_ = div >= 3 ? div : 32;      /* Zeo's default number of divisions [UNVERIFIED-DEP] */ \
      s_->ptype = shape_type; s_->div = dv_; \
      if( shape_type == RKFD_SHAPE_BOX ){ \
        shape_make_box( s_, center[0], bx, by, bz ); \
        memcpy( s_->prm, center[0], sizeof(double)*3 ); s_->prm[3] = bx; s_->prm[4] = by; s_->prm[5] = bz; \
      } else if( shape_type == RKFD_SHAPE_PH && loopf ){ \
        free( s_->vert ); s_->vert = NULL; \
        if( shape_make_prism( s_, loopf, sweep ) < 0 ){ \
          fprintf( stderr, "rkfd: shape %s in %s: malformed loop / prism\n", s_->name, filename ); bad = 1; } \
      } else if( shape_type == RKFD_SHAPE_PH ){ \
        if( shape_build_planes( s_, nface, face ) < 0 ){ \
          fprintf( stderr, "rkfd: shape %s in %s: a face names a vertex that does not exist\n", s_->name, filename ); bad = 1; } \
      } else if( shape_type == RKFD_SHAPE_SPHERE ){ \
        free( s_->vert ); s_->vert = NULL; \
        shape_make_sphere( s_, center[0], radius, dv_ ); \
        memcpy( s_->prm, center[0], sizeof(double)*3 ); s_->prm[3] = radius; \
      } else if( shape_type == RKFD_SHAPE_CYLINDER || shape_type == RKFD_SHAPE_CONE ){ \
        const double *top_ = shape_type == RKFD_SHAPE_CONE ? apex : center[1]; \
        free( s_->vert ); s_->vert = NULL; \
        shape_make_cylinder( s_, center[0], top_, radius, dv_, shape_type == RKFD_SHAPE_CONE ); \
        memcpy( s_->prm, center[0], sizeof(double)*3 ); memcpy( s_->prm+3, top_, sizeof(double)*3 ); s_->prm[6] = radius; \
      } else { s_->nvert = 0; s_->nplane = 0; } \
      cur_shape = -1; \
    } } while(0)

  for( i=0; i<d.nfield; i++ ){
    Field *f = &d.field[i];
    if( !f->key[0] ){ /* section boundary */
      FINISH_SHAPE();
      cur_link = cur_motor = -1; in_init = 0;
      if( strcmp( f->tag, "zeo::shape" ) == 0 ){
        cur_shape = c->nshape++;
        shape_type = 0; nface = 0; vcap = 0; ncenter = 0; div = 0; radius = 0; loopf = NULL; sweep[0] = sweep[1] = sweep[2] = 0;
        memset( center, 0, sizeof(center) ); memset( apex, 0, sizeof(apex) ); bx = by = bz = 0;
      } else if( strcmp( f->tag, "roki::link" ) == 0 ){
        rkfdLinkDesc *l;
        cur_link = c->nlink++;
        l = &c->link[cur_link];
        l->parent = -1; l->motor = -1; l->jtype = RKFD_JOINT_FIXED;
        l->org[0] = l->org[4] = l->org[8] = 1.0;
      } else if( strcmp( f->tag, "roki::motor" ) == 0 ){
        cur_motor = c->nmotor++;
        c->motor[cur_motor].vmax = HUGE_VAL;
        c->motor[cur_motor].vmin = -HUGE_VAL;
        c->motor[cur_motor].gear = 1.0;
      } else if( strcmp( f->tag, "roki::chain::init" ) == 0 ){
        in_init = 1;
      }
      continue;
    }
    if( strcmp( f->tag, "roki::chain" ) == 0 ){
      if( strcmp( f->key, "name" ) == 0 ) sval( f, 0, c->name );
    } else if( cur_shape >= 0 ){
      rkfdShape *s = &c->shape[cur_shape];
      if( strcmp( f->key, "name" ) == 0 ) sval( f, 0, s->name );
      else if( strcmp( f->key, "type" ) == 0 ){
        if( f->nval && strcmp( f->val[0], "box" ) == 0 ) shape_type = RKFD_SHAPE_BOX;
        else if( f->nval && strcmp( f->val[0], "polyhedron" ) == 0 ) shape_type = RKFD_SHAPE_PH;
        else if( f->nval && strcmp( f->val[0], "sphere" ) == 0 ) shape_type = RKFD_SHAPE_SPHERE;
        else if( f->nval && strcmp( f->val[0], "cylinder" ) == 0 ) shape_type = RKFD_SHAPE_CYLINDER;
        else if( f->nval && strcmp( f->val[0], "cone" ) == 0 ) shape_type = RKFD_SHAPE_CONE;
        else {
          shape_type = RKFD_SHAPE_NONE;
          fprintf( stderr, "rkfd: shape %s in %s: type %s is not read (no collision geometry for it)\n", s->name, filename, f->nval ? f->val[0] : "?" );
        }
      }
      else if( strcmp( f->key, "center" ) == 0 ){ if( ncenter < 2 ){ for( k=0; k<3; k++ ) center[ncenter][k] = fval( f, k ); ncenter++; } }
      else if( strcmp( f->key, "radius" ) == 0 ) radius = fval( f, 0 );
      else if( strcmp( f->key, "loop" ) == 0 ) loopf = f;
      else if( strcmp( f->key, "prism" ) == 0 ){ for( k=0; k<3; k++ ) sweep[k] = fval( f, k ); }
      else if( strcmp( f->key, "div" ) == 0 ) div = (int)fval( f, 0 );
      else if( strcmp( f->key, "vert" ) == 0 && shape_type == RKFD_SHAPE_CONE ){ for( k=0; k<3; k++ ) apex[k] = fval( f, k ); }
      else if( strcmp( f->key, "depth" ) == 0 ) bx = fval( f, 0 );
      else if( strcmp( f->key, "width" ) == 0 ) by = fval( f, 0 );
      else if( strcmp( f->key, "height" ) == 0 ) bz = fval( f, 0 );
      else if( strcmp( f->key, "vert" ) == 0 && f->nval >= 4 ){
        int idx = atoi( f->val[0] );
        if( idx < 0 || idx > 1000000 ){
          fprintf( stderr, "rkfd: shape %s in %s: vertex index %d out of range\n", s->name, filename, idx );
          bad = 1; continue;
        }
        if( idx >= vcap ){
          int nc = vcap ? vcap : 16;
          while( nc <= idx ) nc *= 2;
          s->vert = (double *)realloc( s->vert, sizeof(double)*3*nc );
          memset( s->vert + 3*vcap, 0, sizeof(double)*3*(nc-vcap) );
          vcap = nc;
        }
        for( k=0; k<3; k++ ) s->vert[3*idx+k] = fval( f, 1+k );
        if( idx+1 > s->nvert ) s->nvert = idx+1;
      }
      else if( strcmp( f->key, "face" ) == 0 && f->nval >= 3 ){
        if( nface == facecap ){
          facecap = facecap ? 2*facecap : 64;
          face = (int *)realloc( face, sizeof(int)*3*facecap );
        }
        for( k=0; k<3; k++ ) face[3*nface+k] = atoi( f->val[k] );
        nface++;
      }
    } else if( cur_motor >= 0 ){
      rkfdMotor *m = &c->motor[cur_motor];
      if( strcmp( f->key, "name" ) == 0 ) sval( f, 0, m->name );
      else if( strcmp( f->key, "type" ) == 0 ){
        if( f->nval && strcmp( f->val[0], "dc" ) == 0 ) m->type = RKFD_MOTOR_DC;
        else if( f->nval && strcmp( f->val[0], "trq" ) == 0 ) m->type = RKFD_MOTOR_TRQ;
        else m->type = RKFD_MOTOR_NONE;
      }
      else if( strcmp( f->key, "motorconstant" ) == 0 ) m->k = fval( f, 0 );
      else if( strcmp( f->key, "admittance" ) == 0 ) m->admit = fval( f, 0 );
      else if( strcmp( f->key, "maxvoltage" ) == 0 || strcmp( f->key, "max" ) == 0 ) m->vmax = fval( f, 0 );
      else if( strcmp( f->key, "minvoltage" ) == 0 || strcmp( f->key, "min" ) == 0 ) m->vmin = fval( f, 0 );
      else if( strcmp( f->key, "gearratio" ) == 0 ) m->gear = fval( f, 0 );
      else if( strcmp( f->key, "rotorinertia" ) == 0 ) m->rotor_inertia = fval( f, 0 );
      else if( strcmp( f->key, "gearinertia" ) == 0 ) m->gear_inertia = fval( f, 0 );
    } else if( cur_link >= 0 ){
      rkfdLinkDesc *l = &c->link[cur_link];
      if( strcmp( f->key, "name" ) == 0 ) sval( f, 0, l->name );
      else if( strcmp( f->key, "jointtype" ) == 0 ){
        l->jtype = f->nval ? parse_jointtype( f->val[0] ) : -1;
        if( l->jtype < 0 ){
          fprintf( stderr, "rkfd: unsupported jointtype %s in %s\n", f->nval ? f->val[0] : "?", filename );
          free( face ); doc_free( &d ); rkfdChainDescFree( c );
          return NULL;
        }
      }
      else if( strcmp( f->key, "mass" ) == 0 ) l->mass = fval( f, 0 );
      else if( strcmp( f->key, "stuff" ) == 0 ) sval( f, 0, l->stuff );
      else if( strcmp( f->key, "COM" ) == 0 ){
        if( f->nval && strcmp( f->val[0], "auto" ) == 0 ) l->auto_com = 1; else for( k=0; k<3; k++ ) l->com[k] = fval( f, k );
      }
      else if( strcmp( f->key, "inertia" ) == 0 ){
        if( f->nval && strcmp( f->val[0], "auto" ) == 0 ) l->auto_inertia = 1; else for( k=0; k<9; k++ ) l->inertia[k] = fval( f, k );
      }
      else if( strcmp( f->key, "forcethreshold" ) == 0 ) l->ep_f = fval( f, 0 );
      else if( strcmp( f->key, "torquethreshold" ) == 0 ) l->ep_t = fval( f, 0 );
      else if( strcmp( f->key, "frame" ) == 0 ){
        for( j=0; j<3; j++ ){
          for( k=0; k<3; k++ ) l->org[3*j+k] = fval( f, 4*j+k );
          l->org[9+j] = fval( f, 4*j+3 );
        }
      }
      else if( strcmp( f->key, "pos" ) == 0 ){ for( k=0; k<3; k++ ) l->org[9+k] = fval( f, k ); }
      else if( strcmp( f->key, "att" ) == 0 ){ for( k=0; k<9; k++ ) l->org[k] = fval( f, k ); }
      else if( strcmp( f->key, "parent" ) == 0 ) sval( f, 0, l->parent_name );
      else if( strcmp( f->key, "motor" ) == 0 ) sval( f, 0, l->motor_name );
      else if( strcmp( f->key, "stiffness" ) == 0 ) l->stiff = fval( f, 0 );
      else if( strcmp( f->key, "viscosity" ) == 0 ) l->visc = fval( f, 0 );
      else if( strcmp( f->key, "coulomb" ) == 0 ) l->coulomb = fval( f, 0 );
      else if( strcmp( f->key, "staticfriction" ) == 0 ) l->sfric = fval( f, 0 );
      else if( strcmp( f->key, "shape" ) == 0 && f->nval ){
        int si = find_name( f->val[0], (const char *)c->shape, sizeof(rkfdShape), c->nshape );
        if( si >= 0 && l->nshape < 8 ) l->shape[l->nshape++] = si;
      }
    }
    (void)in_init;
  }
  FINISH_SHAPE();
#undef FINISH_SHAPE
  free( face );
  if( bad ){ doc_free( &d ); rkfdChainDescFree( c ); return NULL; }
  /* "COM: auto" / "inertia: auto": uniform density over the link's shapes */
  for( i=0; i<c->nlink; i++ ){
    rkfdLinkDesc *l = &c->link[i];
    double vol = 0, first[3] = {0,0,0}, second[6] = {0,0,0,0,0,0}, rho, cm[3];
    if( !l->auto_com && !l->auto_inertia ) continue;
    for( j=0; j<l->nshape; j++ ){
      double v, f1[3], s2[6];
      shape_moments( &c->shape[l->shape[j]], &v, f1, s2 );
      vol += v; for( k=0; k<3; k++ ) first[k] += f1[k]; for( k=0; k<6; k++ ) second[k] += s2[k];
    }
    if( vol < 1e-18 ){
      fprintf( stderr, "rkfd: link %s in %s: COM / inertia `auto` without a shape that has volume\n", l->name, filename );
      doc_free( &d ); rkfdChainDescFree( c ); return NULL;
    }
    rho = l->mass/vol;
    for( k=0; k<3; k++ ) cm[k] = first[k]/vol;
    if( l->auto_com ) memcpy( l->com, cm, sizeof(cm) );
    if( l->auto_inertia ){
      /* about the COM used for the link: second moments shifted by the parallel-axis terms */
      const double *g = l->com;
      const double xx = rho*second[0] - l->mass*( 2*g[0]*cm[0] - g[0]*g[0] ), yy = rho*second[1] - l->mass*( 2*g[1]*cm[1] - g[1]*g[1] ), zz = rho*second[2] - l->mass*( 2*g[2]*cm[2] - g[2]*g[2] );
      const double xy = rho*second[3] - l->mass*( g[0]*cm[1] + g[1]*cm[0] - g[0]*g[1] ), yz = rho*second[4] - l->mass*( g[1]*cm[2] + g[2]*cm[1] - g[1]*g[2] ), zx = rho*second[5] - l->mass*( g[2]*cm[0] + g[0]*cm[2] - g[2]*g[0] );
      l->inertia[0] = yy+zz; l->inertia[4] = zz+xx; l->inertia[8] = xx+yy;
      l->inertia[1] = l->inertia[3] = -xy; l->inertia[5] = l->inertia[7] = -yz; l->inertia[2] = l->inertia[6] = -zx;
    }
  }

  /* resolve parents and motors */
  for( i=0; i<c->nlink; i++ ){
    rkfdLinkDesc *l = &c->link[i];
    if( l->parent_name[0] ){
      l->parent = find_name( l->parent_name, (const char *)c->link, sizeof(rkfdLinkDesc), c->nlink );
      if( l->parent < 0 ){
        fprintf( stderr, "rkfd: unknown parent %s of link %s\n", l->parent_name, l->name );
        doc_free( &d ); rkfdChainDescFree( c );
        return NULL;
      }
    }
    if( l->motor_name[0] )
      l->motor = find_name( l->motor_name, (const char *)c->motor, sizeof(rkfdMotor), c->nmotor );
  }
  /* stable topological order so that parent index < child index */
  order  = (int *)malloc( sizeof(int)*c->nlink );
  newidx = (int *)malloc( sizeof(int)*c->nlink );
  for( i=0; i<c->nlink; i++ ) newidx[i] = -1;
  k = 0;
  while( k < c->nlink ){
    int progressed = 0;
    for( i=0; i<c->nlink; i++ ){
      if( newidx[i] >= 0 ) continue;
      if( c->link[i].parent < 0 || newidx[c->link[i].parent] >= 0 ){
        newidx[i] = k; order[k++] = i; progressed = 1;
      }
    }
    if( !progressed ){
      fprintf( stderr, "rkfd: cyclic parent relation in %s\n", filename );
      free( order ); free( newidx ); doc_free( &d ); rkfdChainDescFree( c );
      return NULL;
    }
  }
  sorted = (rkfdLinkDesc *)malloc( sizeof(rkfdLinkDesc)*c->nlink );
  for( k=0; k<c->nlink; k++ ){
    sorted[k] = c->link[order[k]];
    if( sorted[k].parent >= 0 ) sorted[k].parent = newidx[sorted[k].parent];
  }
  free( c->link ); c->link = sorted;
  free( order ); free( newidx );

  c->ndof = 0;
  for( i=0; i<c->nlink; i++ ) c->ndof += rkfd_joint_dof( c->link[i].jtype );
  c->init_dis = (double *)calloc( c->ndof ? c->ndof : 1, sizeof(double) );

  /* [roki::chain::init] */
  {
    int in = 0;
    for( i=0; i<d.nfield; i++ ){
      Field *f = &d.field[i];
      if( !f->key[0] ){ in = ( strcmp( f->tag, "roki::chain::init" ) == 0 ); continue; }
      if( !in ) continue;
      if( strcmp( f->key, "joint" ) == 0 && f->nval >= 1 ){
        int li = find_name( f->val[0], (const char *)c->link, sizeof(rkfdLinkDesc), c->nlink );
        int off = 0, nd;
        if( li < 0 ) continue;
        for( j=0; j<li; j++ ) off += rkfd_joint_dof( c->link[j].jtype );
        nd = rkfd_joint_dof( c->link[li].jtype );
        for( k=0; k<nd; k++ ) c->init_dis[off+k] = fval( f, 1+k );
      } else if( strcmp( f->key, "frame" ) == 0 && f->nval >= 12 ){
        /* chain base frame: composed onto the root link's org frame */
        double R[9], p[3], Ro[9], po[3];
        rkfdLinkDesc *l = &c->link[0];
        for( j=0; j<3; j++ ){ for( k=0; k<3; k++ ) R[3*j+k] = fval( f, 4*j+k ); p[j] = fval( f, 4*j+3 ); }
        memcpy( Ro, l->org, sizeof(Ro) ); memcpy( po, l->org+9, sizeof(po) );
        for( j=0; j<3; j++ ){
          for( k=0; k<3; k++ ) l->org[3*j+k] = R[3*j]*Ro[k] + R[3*j+1]*Ro[3+k] + R[3*j+2]*Ro[6+k];
          l->org[9+j] = p[j] + R[3*j]*po[0] + R[3*j+1]*po[1] + R[3*j+2]*po[2];
        }
      }
    }
  }
  doc_free( &d );
  return c;
}

void rkfdChainDescFree(rkfdChainDesc *c)
{
  int i;
  if( !c ) return;
  for( i=0; i<c->nshape; i++ ){ free( c->shape[i].vert ); free( c->shape[i].plane ); free( c->shape[i].face ); }
  free( c->shape ); free( c->motor ); free( c->link ); free( c->init_dis );
  free( c );
}

/* ------------------------------------------------------------------------ */
/* writer: the role of rkChainFPrintZTK (reference src/rkfd_sim.c:587-593, rkFDPrint) */
void rkfdChainWriteZTK(FILE *fp, const rkfdChainDesc *c, const double *dis)
{
  static const char *jname[] = { "fixed", "revolute", "prismatic", "float", "spherical", "breakablefloat" };
  int i, j, k, off = 0;
  fprintf( fp, "[roki::chain]\nname : %s\n\n", c->name );
  for( i=0; i<c->nshape; i++ ){
    const rkfdShape *s = &c->shape[i];
    if( s->ptype == RKFD_SHAPE_NONE ) continue;
    fprintf( fp, "[zeo::shape]\nname: %s\n", s->name );
    switch( s->ptype ){
    case RKFD_SHAPE_BOX:
      fprintf( fp, "type: box\ncenter: { %.17g, %.17g, %.17g }\ndepth: %.17g\nwidth: %.17g\nheight: %.17g\n", s->prm[0], s->prm[1], s->prm[2], s->prm[3], s->prm[4], s->prm[5] ); break;
    case RKFD_SHAPE_SPHERE:
      fprintf( fp, "type: sphere\ncenter: { %.17g, %.17g, %.17g }\nradius: %.17g\ndiv: %d\n", s->prm[0], s->prm[1], s->prm[2], s->prm[3], s->div ); break;
    case RKFD_SHAPE_CYLINDER:
      fprintf( fp, "type: cylinder\ncenter: { %.17g, %.17g, %.17g }\ncenter: { %.17g, %.17g, %.17g }\nradius: %.17g\ndiv: %d\n",
               s->prm[0], s->prm[1], s->prm[2], s->prm[3], s->prm[4], s->prm[5], s->prm[6], s->div ); break;
    case RKFD_SHAPE_CONE:
      fprintf( fp, "type: cone\ncenter: { %.17g, %.17g, %.17g }\nvert: { %.17g, %.17g, %.17g }\nradius: %.17g\ndiv: %d\n",
               s->prm[0], s->prm[1], s->prm[2], s->prm[3], s->prm[4], s->prm[5], s->prm[6], s->div ); break;
    default:
      fprintf( fp, "type: polyhedron\n" );
      for( j=0; j<s->nvert; j++ ) fprintf( fp, "vert: %d { %.17g, %.17g, %.17g }\n", j, s->vert[3*j], s->vert[3*j+1], s->vert[3*j+2] );
      for( j=0; j<s->nface; j++ ) fprintf( fp, "face: %d %d %d\n", s->face[3*j], s->face[3*j+1], s->face[3*j+2] );
    }
    fprintf( fp, "\n" );
  }
  for( i=0; i<c->nmotor; i++ ){
    const rkfdMotor *m = &c->motor[i];
    fprintf( fp, "[roki::motor]\nname : %s\n", m->name );
    if( m->type == RKFD_MOTOR_DC )
      fprintf( fp, "type: dc\nmotorconstant : %.17g\nadmittance : %.17g\nmaxvoltage : %.17g\nminvoltage : %.17g\ngearratio : %.17g\nrotorinertia : %.17g\ngearinertia : %.17g\n\n",
               m->k, m->admit, m->vmax, m->vmin, m->gear, m->rotor_inertia, m->gear_inertia );
    else if( m->type == RKFD_MOTOR_TRQ ){
      fprintf( fp, "type: trq\n" );
      if( m->vmax < HUGE_VAL ) fprintf( fp, "max: %.17g\n", m->vmax );
      if( m->vmin > -HUGE_VAL ) fprintf( fp, "min: %.17g\n", m->vmin );
      fprintf( fp, "\n" );
    } else fprintf( fp, "type: none\n\n" );
  }
  for( i=0; i<c->nlink; i++ ){
    const rkfdLinkDesc *l = &c->link[i];
    fprintf( fp, "[roki::link]\nname: %s\njointtype: %s\n", l->name, jname[l->jtype] );
    if( l->jtype == RKFD_JOINT_REVOL || l->jtype == RKFD_JOINT_PRISM ){
      fprintf( fp, "stiffness: %.17g\nviscosity: %.17g\ncoulomb: %.17g\nstaticfriction: %.17g\n", l->stiff, l->visc, l->coulomb, l->sfric );
      if( l->motor >= 0 ) fprintf( fp, "motor: %s\n", c->motor[l->motor].name );
    }
    if( l->jtype == RKFD_JOINT_BRFLOAT ) fprintf( fp, "forcethreshold: %.17g\ntorquethreshold: %.17g\n", l->ep_f, l->ep_t );
    fprintf( fp, "mass: %.17g\n", l->mass );
    if( l->stuff[0] ) fprintf( fp, "stuff: %s\n", l->stuff );
    fprintf( fp, "COM: { %.17g, %.17g, %.17g }\ninertia: {\n", l->com[0], l->com[1], l->com[2] );
    for( j=0; j<3; j++ ) fprintf( fp, " %.17g, %.17g, %.17g\n", l->inertia[3*j], l->inertia[3*j+1], l->inertia[3*j+2] );
    fprintf( fp, "}\nframe: {\n" );
    for( j=0; j<3; j++ ) fprintf( fp, " %.17g, %.17g, %.17g, %.17g\n", l->org[3*j], l->org[3*j+1], l->org[3*j+2], l->org[9+j] );
    fprintf( fp, "}\n" );
    for( j=0; j<l->nshape; j++ ) fprintf( fp, "shape: %s\n", c->shape[l->shape[j]].name );
    if( l->parent >= 0 ) fprintf( fp, "parent: %s\n", c->link[l->parent].name );
    fprintf( fp, "\n" );
  }
  fprintf( fp, "[roki::chain::init]\n" );
  for( i=0; i<c->nlink; i++ ){
    const int nd = rkfd_joint_dof( c->link[i].jtype );
    if( nd > 0 ){
      fprintf( fp, "joint: %s", c->link[i].name );
      for( k=0; k<nd; k++ ) fprintf( fp, " %.17g", dis ? dis[off+k] : c->init_dis[off+k] );
      fprintf( fp, "\n" );
    }
    off += nd;
  }
  fprintf( fp, "\n" );
}

int rkfdContactInfoReadZTK(const char *filename, rkfdContactInfo **out)
{
  Doc d;
  int i, n = 0, cur = -1, cnt = 0;
  rkfdContactInfo *ci;

  *out = NULL;
  if( doc_read( &d, filename ) < 0 ) return -1;
  for( i=0; i<d.nfield; i++ )
    if( !d.field[i].key[0] && strcmp( d.field[i].tag, "roki::contact" ) == 0 ) n++;
  ci = (rkfdContactInfo *)calloc( n ? n : 1, sizeof(rkfdContactInfo) );
  for( i=0; i<d.nfield; i++ ){
    Field *f = &d.field[i];
    if( !f->key[0] ){
      cur = strcmp( f->tag, "roki::contact" ) == 0 ? cnt++ : -1;
      continue;
    }
    if( cur < 0 || strcmp( f->tag, "roki::contact" ) != 0 ) continue;
    if( strcmp( f->key, "bind" ) == 0 ){ sval( f, 0, ci[cur].stuff[0] ); sval( f, 1, ci[cur].stuff[1] ); }
    else if( strcmp( f->key, "staticfriction" ) == 0 ) ci[cur].sf = fval( f, 0 );
    else if( strcmp( f->key, "kineticfriction" ) == 0 ) ci[cur].kf = fval( f, 0 );
    else if( strcmp( f->key, "compensation" ) == 0 ){ ci[cur].k = fval( f, 0 ); ci[cur].type = RKFD_CONTACT_RIGID; }
    else if( strcmp( f->key, "relaxation" ) == 0 ){ ci[cur].l = fval( f, 0 ); ci[cur].type = RKFD_CONTACT_RIGID; }
    else if( strcmp( f->key, "elasticity" ) == 0 ){ ci[cur].e = fval( f, 0 ); ci[cur].type = RKFD_CONTACT_ELASTIC; }
    else if( strcmp( f->key, "viscosity" ) == 0 ){ ci[cur].v = fval( f, 0 ); ci[cur].type = RKFD_CONTACT_ELASTIC; }
  }
  doc_free( &d );
  *out = ci;
  return n;
}

static void *dup_mem(const void *p, size_t n)
{
  void *q;
  if( !p || n == 0 ) return NULL;
  if( ( q = malloc( n ) ) ) memcpy( q, p, n );
  return q;
}

rkfdChainDesc *rkfdChainDescClone(const rkfdChainDesc *c)
{
  rkfdChainDesc *d;
  int i;
  if( !c || !( d = (rkfdChainDesc *)calloc( 1, sizeof(rkfdChainDesc) ) ) ) return NULL;
  *d = *c;
  d->link = (rkfdLinkDesc *)dup_mem( c->link, sizeof(rkfdLinkDesc)*c->nlink );
  d->motor = (rkfdMotor *)dup_mem( c->motor, sizeof(rkfdMotor)*c->nmotor );
  d->init_dis = (double *)dup_mem( c->init_dis, sizeof(double)*c->ndof );
  d->shape = (rkfdShape *)dup_mem( c->shape, sizeof(rkfdShape)*c->nshape );
  for( i=0; d->shape && i<c->nshape; i++ ){
    d->shape[i].vert = (double *)dup_mem( c->shape[i].vert, sizeof(double)*3*c->shape[i].nvert );
    d->shape[i].plane = (double *)dup_mem( c->shape[i].plane, sizeof(double)*4*c->shape[i].nplane );
    d->shape[i].face = (int *)dup_mem( c->shape[i].face, sizeof(int)*3*c->shape[i].nface );
  }
  if( ( c->nlink && !d->link ) || ( c->nmotor && !d->motor ) || ( c->nshape && !d->shape ) || ( c->ndof && !d->init_dis ) ){
    rkfdChainDescFree( d );
    return NULL;
  }
  return d;
}
