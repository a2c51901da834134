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

/* build the deduplicated outward face planes of a convex polyhedron from its triangles */
static void shape_build_planes(rkfdShape *s, int nface, const int *face)
{
  int i, j, k;
  double cen[3] = {0,0,0}, e1[3], e2[3], n[4], len;

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
  int cur_shape = -1, shape_type = 0, nface = 0, facecap = 0, *face = NULL, vcap = 0;
  double center[3] = {0,0,0}, bx = 0, by = 0, bz = 0;
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
      if( shape_type == 1 ) shape_make_box( s_, center, bx, by, bz ); \
      else if( shape_type == 2 ) shape_build_planes( s_, nface, face ); \
      else { s_->nvert = 0; s_->nplane = 0; } \
      cur_shape = -1; \
    } } while(0)

  for( i=0; i<d.nfield; i++ ){
    Field *f = &d.field[i];
    if( !f->key[0] ){ /* section boundary */
      FINISH_SHAPE();
      cur_link = cur_motor = -1; in_init = 0;
      if( strcmp( f->tag, "zeo::shape" ) == 0 ){
        cur_shape = c->nshape++;
        shape_type = 0; nface = 0; vcap = 0;
        center[0] = center[1] = center[2] = 0; bx = by = bz = 0;
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
        if( f->nval && strcmp( f->val[0], "box" ) == 0 ) shape_type = 1;
        else if( f->nval && strcmp( f->val[0], "polyhedron" ) == 0 ) shape_type = 2;
        else shape_type = 0; /* curved primitives: skipped */
      }
      else if( strcmp( f->key, "center" ) == 0 ){ for( k=0; k<3; k++ ) center[k] = fval( f, k ); }
      else if( strcmp( f->key, "depth" ) == 0 ) bx = fval( f, 0 );
      else if( strcmp( f->key, "width" ) == 0 ) by = fval( f, 0 );
      else if( strcmp( f->key, "height" ) == 0 ) bz = fval( f, 0 );
      else if( strcmp( f->key, "vert" ) == 0 && f->nval >= 4 ){
        int idx = atoi( f->val[0] );
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
      else if( strcmp( f->key, "COM" ) == 0 ){ for( k=0; k<3; k++ ) l->com[k] = fval( f, k ); }
      else if( strcmp( f->key, "inertia" ) == 0 ){ for( k=0; k<9; k++ ) l->inertia[k] = fval( f, k ); }
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
  for( i=0; i<c->nshape; i++ ){ free( c->shape[i].vert ); free( c->shape[i].plane ); }
  free( c->shape ); free( c->motor ); free( c->link ); free( c->init_dis );
  free( c );
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
  }
  if( ( c->nlink && !d->link ) || ( c->nmotor && !d->motor ) || ( c->nshape && !d->shape ) || ( c->ndof && !d->init_dis ) ){
    rkfdChainDescFree( d );
    return NULL;
  }
  return d;
}
