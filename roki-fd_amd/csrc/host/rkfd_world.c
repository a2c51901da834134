/* rkfd_world.c - flatten registered chains into an rkfdModel.
 * Role in the reference: rkFD cell list + packed joint state offsets
 * (reference src/rkfd_sim.c:72-110,188-209), rkCD cell/pair registry (RoKi rk_cd,
 * un-vendored) and contact-info association by "stuff" (reference src/rkfd_sim.c:200-207).
 */
#include <stdlib.h>
#include <string.h>
#include "rkfd_world.h"

void rkfdWorldInit(rkfdWorld *w)
{
  memset( w, 0, sizeof(rkfdWorld) );
  /* default contact info of the default solver (Vert), reference src/rkfd_vert.c:340-348 */
  w->cidef.type = RKFD_CONTACT_RIGID;
  w->cidef.k = 1000.0; w->cidef.l = 1.0; w->cidef.sf = 0.5; w->cidef.kf = 0.3;
  /* rkFDPrpInit defaults, reference src/rkfd_property.c:10-18, include/roki_fd/rkfd_defs.h:15-23 */
  w->model.dt = 0.001;
  w->model.friction_weight = 100.0;
  w->model.max_iter = 10;
  w->model.solver = RKFD_SOLVER_VERT;
  w->model.pyramid = 8;
}

void rkfdWorldDestroy(rkfdWorld *w)
{
  int i;
  for( i=0; i<w->nchain; i++ ) rkfdChainDescFree( w->chain[i] );
  free( w->chain ); free( w->noself ); free( w->ci ); free( w->blob );
  memset( w, 0, sizeof(rkfdWorld) );
}

int rkfdWorldAddChain(rkfdWorld *w, rkfdChainDesc *c)
{
  int n = w->nchain;

  w->chain = (rkfdChainDesc **)realloc( w->chain, sizeof(rkfdChainDesc*)*(n+1) );
  w->noself = (unsigned char *)realloc( w->noself, (size_t)(n+1) );
  if( !w->chain || !w->noself ) return -1;
  w->noself[n] = 0;
  w->chain[n] = c;
  w->nchain = n+1;
  w->built = 0;
  return n;
}

void rkfdWorldRemoveChain(rkfdWorld *w, int chain)
{
  int n = w->nchain, i;
  if( chain < 0 || chain >= n ) return;
  rkfdChainDescFree( w->chain[chain] );
  for( i=chain; i<n-1; i++ ){ w->chain[i] = w->chain[i+1]; w->noself[i] = w->noself[i+1]; }
  w->nchain = n-1;
  w->built = 0;
}

void rkfdWorldPairChainUnreg(rkfdWorld *w, int chain)
{
  if( chain < 0 || chain >= w->nchain ) return;
  w->noself[chain] = 1;
  w->built = 0;
}

int rkfdWorldSetContactInfo(rkfdWorld *w, const char *filename)
{
  rkfdContactInfo *ci;
  int n = rkfdContactInfoReadZTK( filename, &ci );
  if( n < 0 ) return -1;
  free( w->ci );
  w->ci = ci; w->nci = n;
  w->built = 0;
  return 0;
}

int rkfdWorldChainDofOffset(const rkfdWorld *w, int chain)
{
  int i, off = 0;
  for( i=0; i<chain; i++ ) off += w->chain[i]->ndof;
  return off;
}

int rkfdWorldChainLinkOffset(const rkfdWorld *w, int chain)
{
  int i, off = 0;
  for( i=0; i<chain; i++ ) off += w->chain[i]->nlink;
  return off;
}

static int assoc_ci(const rkfdWorld *w, const char *s0, const char *s1)
{
  int i;
  for( i=0; i<w->nci; i++ ){
    if( ( strcmp( w->ci[i].stuff[0], s0 ) == 0 && strcmp( w->ci[i].stuff[1], s1 ) == 0 ) ||
        ( strcmp( w->ci[i].stuff[0], s1 ) == 0 && strcmp( w->ci[i].stuff[1], s0 ) == 0 ) ) return i;
  }
  return w->nci; /* default entry */
}

/* The link a link is rigidly attached to: itself unless it hangs on a FIXED joint, then what its parent is attached to
 * (-1: the world).  Two cells of one chain whose links share it can never move against each other. */
static int rigid_rep(const rkfdChainDesc *cd, int link)
{
  while( link >= 0 && cd->link[link].jtype == RKFD_JOINT_FIXED ) link = cd->link[link].parent;
  return link;
}

/* Does registration pair these two cells (chain, chain-local link)?  rkCDChainReg -> rkCDPairReg [RoKi, UNVERIFIED-DEP]:
 * every new cell against every cell registered before it, of another chain or of its own (the reference's drivers then drop
 * an articulated chain's own pairs with rkCDPairChainUnreg, example/chain/arm_box_test.c:49; arm_wall_test.c keeps the
 * wall's: its bricks are links of one chain and do collide).  Not paired: two cells on the same link, and - this build -
 * two cells of one chain that are rigidly attached to each other (only FIXED joints between them): such a pair cannot
 * change any acceleration, it would only occupy contact slots (RoKi skips the static-static case the same way; DEVIATIONS.md). */
static int cells_pair(const rkfdWorld *w, int cx, int lx, int cy, int ly)
{
  if( cx != cy ) return 1;
  if( w->noself[cx] || lx == ly ) return 0;
  return rigid_rep( w->chain[cx], lx ) != rigid_rep( w->chain[cx], ly );
}

/* bump allocator over one blob */
typedef struct { char *base; size_t off; } Arena;
static void *arena_get(Arena *a, size_t bytes)
{
  void *p;
  a->off = ( a->off + 15 ) & ~(size_t)15;
  p = a->base ? a->base + a->off : NULL;
  a->off += bytes;
  return p;
}

int rkfdWorldBuild(rkfdWorld *w)
{
  int pass, c, i, k, s;
  int nlink = 0, ndof = 0, nshape = 0, nvert = 0, nplane = 0, npair = 0, ncand = 0;
  rkfdModel *m = &w->model;
  Arena a = { NULL, 0 };
  /* arrays (non-const views while filling) */
  int *parent = NULL, *jtype = NULL, *dofoff = NULL, *chain = NULL, *mtype = NULL;
  double *org = NULL, *mass = NULL, *com = NULL, *inertia = NULL;
  double *stiff = NULL, *visc = NULL, *coulomb = NULL, *sfric = NULL;
  double *mot_k = NULL, *mot_admit = NULL, *mot_vmax = NULL, *mot_vmin = NULL, *mot_gear = NULL, *mot_inertia = NULL;
  double *brk_f = NULL, *brk_t = NULL;
  int *shape_link = NULL, *shape_voff = NULL, *shape_foff = NULL, *shape_chain = NULL, *shape_convex = NULL;
  int *shape_slide_mode = NULL; double *shape_slide_vel = NULL, *shape_slide_axis = NULL;
  double *verts = NULL, *planes = NULL;
  int *pair_shape = NULL, *pair_ci = NULL, *ci_type = NULL;
  double *ci_sf = NULL, *ci_kf = NULL, *ci_k = NULL, *ci_l = NULL, *ci_e = NULL, *ci_v = NULL;
  int *cand_pair = NULL, *cand_side = NULL, *cand_vert = NULL;
  const char **shape_stuff = NULL;

  /* counts */
  for( c=0; c<w->nchain; c++ ){
    rkfdChainDesc *cd = w->chain[c];
    nlink += cd->nlink; ndof += cd->ndof;
    for( i=0; i<cd->nlink; i++ )
      for( s=0; s<cd->link[i].nshape; s++ ){
        rkfdShape *sh = &cd->shape[cd->link[i].shape[s]];
        if( sh->nvert == 0 || sh->nplane == 0 ) continue;
        nshape++; nvert += sh->nvert; nplane += sh->nplane;
      }
  }

  /* count pairs and candidates */
  {
    int *sc = (int *)malloc( sizeof(int)*( nshape ? nshape : 1 ) );
    int *sl = (int *)malloc( sizeof(int)*( nshape ? nshape : 1 ) );
    int *sv = (int *)malloc( sizeof(int)*( nshape ? nshape : 1 ) );
    int *scv = (int *)malloc( sizeof(int)*( nshape ? nshape : 1 ) );
    int n = 0, x, y;
    for( c=0; c<w->nchain; c++ ){
      rkfdChainDesc *cd = w->chain[c];
      for( i=0; i<cd->nlink; i++ )
        for( s=0; s<cd->link[i].nshape; s++ ){
          rkfdShape *sh = &cd->shape[cd->link[i].shape[s]];
          if( sh->nvert == 0 || sh->nplane == 0 ) continue;
          sc[n] = c; sl[n] = i; sv[n] = sh->nvert; scv[n] = sh->convex; n++;
        }
    }
    for( y=0; y<n; y++ )
      for( x=0; x<y; x++ ){
        if( !cells_pair( w, sc[x], sl[x], sc[y], sl[y] ) ) continue;
        /* a shape's vertices are candidates against the OTHER shape of the pair only when that one is convex (the
         * inside test is the intersection of its face half-spaces) */
        npair++; ncand += ( scv[y] ? sv[x] : 0 ) + ( scv[x] ? sv[y] : 0 );
      }
    free( sc ); free( sl ); free( sv ); free( scv );
  }

  for( pass=0; pass<2; pass++ ){
    int li = 0, si = 0, vi = 0, fi = 0, doff = 0, pi = 0, ci = 0;
    if( pass == 1 ){
      free( w->blob );
      w->blob = calloc( 1, a.off + 64 );
      if( !w->blob ) return -1;
      a.base = (char *)w->blob; a.off = 0;
    }
#define GET(ptr,type,n) ptr = (type *)arena_get( &a, sizeof(type)*( (n) > 0 ? (n) : 1 ) )
    GET( parent, int, nlink ); GET( jtype, int, nlink ); GET( dofoff, int, nlink ); GET( chain, int, nlink );
    GET( mtype, int, nlink );
    GET( org, double, nlink*12 ); GET( mass, double, nlink ); GET( com, double, nlink*3 ); GET( inertia, double, nlink*9 );
    GET( stiff, double, nlink ); GET( visc, double, nlink ); GET( coulomb, double, nlink ); GET( sfric, double, nlink );
    GET( mot_k, double, nlink ); GET( mot_admit, double, nlink ); GET( mot_vmax, double, nlink );
    GET( mot_vmin, double, nlink ); GET( mot_gear, double, nlink ); GET( mot_inertia, double, nlink );
    GET( brk_f, double, nlink ); GET( brk_t, double, nlink );
    GET( shape_link, int, nshape ); GET( shape_voff, int, nshape+1 ); GET( shape_foff, int, nshape+1 );
    GET( shape_chain, int, nshape ); GET( shape_convex, int, nshape );
    GET( shape_slide_mode, int, nshape ); GET( shape_slide_vel, double, nshape ); GET( shape_slide_axis, double, nshape*3 );
    GET( shape_stuff, const char *, nshape );
    GET( verts, double, nvert*3 ); GET( planes, double, nplane*4 );
    GET( pair_shape, int, npair*2 ); GET( pair_ci, int, npair );
    GET( ci_type, int, w->nci+1 ); GET( ci_sf, double, w->nci+1 ); GET( ci_kf, double, w->nci+1 );
    GET( ci_k, double, w->nci+1 ); GET( ci_l, double, w->nci+1 ); GET( ci_e, double, w->nci+1 ); GET( ci_v, double, w->nci+1 );
    GET( cand_pair, int, ncand ); GET( cand_side, int, ncand ); GET( cand_vert, int, ncand );
#undef GET
    if( pass == 0 ) continue;
    /* fill */
    for( c=0; c<w->nchain; c++ ){
      rkfdChainDesc *cd = w->chain[c];
      int lbase = li;
      for( i=0; i<cd->nlink; i++, li++ ){
        rkfdLinkDesc *l = &cd->link[i];
        parent[li] = l->parent < 0 ? -1 : lbase + l->parent;
        jtype[li] = l->jtype; dofoff[li] = doff; chain[li] = c;
        doff += rkfd_joint_dof( l->jtype );
        memcpy( &org[12*li], l->org, sizeof(double)*12 );
        mass[li] = l->mass;
        memcpy( &com[3*li], l->com, sizeof(double)*3 );
        memcpy( &inertia[9*li], l->inertia, sizeof(double)*9 );
        if( l->jtype == RKFD_JOINT_BRFLOAT ){ brk_f[li] = l->ep_f; brk_t[li] = l->ep_t; }
        if( rkfd_joint_dof( l->jtype ) == 1 ){
          stiff[li] = l->stiff; visc[li] = l->visc; coulomb[li] = l->coulomb; sfric[li] = l->sfric;
          if( l->motor >= 0 ){
            rkfdMotor *mo = &cd->motor[l->motor];
            mtype[li] = mo->type;
            mot_k[li] = mo->k; mot_admit[li] = mo->admit; mot_vmax[li] = mo->vmax; mot_vmin[li] = mo->vmin;
            mot_gear[li] = mo->gear; mot_inertia[li] = mo->rotor_inertia + mo->gear_inertia;
          }
        }
        for( s=0; s<l->nshape; s++ ){
          rkfdShape *sh = &cd->shape[l->shape[s]];
          if( sh->nvert == 0 || sh->nplane == 0 ) continue;
          shape_link[si] = li; shape_chain[si] = c; shape_stuff[si] = l->stuff; shape_convex[si] = sh->convex;
          shape_voff[si] = vi; shape_foff[si] = fi;
          shape_slide_mode[si] = sh->slide_mode; shape_slide_vel[si] = sh->slide_vel;
          memcpy( &shape_slide_axis[3*si], sh->slide_axis, sizeof(double)*3 );
          memcpy( &verts[3*vi], sh->vert, sizeof(double)*3*sh->nvert );
          memcpy( &planes[4*fi], sh->plane, sizeof(double)*4*sh->nplane );
          vi += sh->nvert; fi += sh->nplane; si++;
        }
      }
    }
    shape_voff[nshape] = vi; shape_foff[nshape] = fi;
    /* contact infos: file entries then the solver default */
    for( i=0; i<=w->nci; i++ ){
      const rkfdContactInfo *q = i < w->nci ? &w->ci[i] : &w->cidef;
      ci_type[i] = q->type; ci_sf[i] = q->sf; ci_kf[i] = q->kf;
      ci_k[i] = q->k; ci_l[i] = q->l; ci_e[i] = q->e; ci_v[i] = q->v;
    }
    /* pairs: each later-registered cell against every earlier cell (cells_pair) */
    {
      int x, y;
      for( y=0; y<nshape; y++ )
        for( x=0; x<y; x++ ){
          if( !cells_pair( w, shape_chain[x], shape_link[x] - rkfdWorldChainLinkOffset( w, shape_chain[x] ),
                              shape_chain[y], shape_link[y] - rkfdWorldChainLinkOffset( w, shape_chain[y] ) ) ) continue;
          pair_shape[2*pi] = x; pair_shape[2*pi+1] = y;
          pair_ci[pi] = assoc_ci( w, shape_stuff[x], shape_stuff[y] );
          for( s=0; s<2; s++ ){
            int sh = pair_shape[2*pi+s];
            if( !shape_convex[pair_shape[2*pi+1-s]] ) continue;
            for( k=shape_voff[sh]; k<shape_voff[sh+1]; k++ ){
              cand_pair[ci] = pi; cand_side[ci] = s; cand_vert[ci] = k; ci++;
            }
          }
          pi++;
        }
    }
  }

  m->nlink = nlink; m->ndof = ndof; m->nchain = w->nchain;
  m->parent = parent; m->jtype = jtype; m->dofoff = dofoff; m->chain = chain;
  m->org = org; m->mass = mass; m->com = com; m->inertia = inertia;
  m->stiff = stiff; m->visc = visc; m->coulomb = coulomb; m->sfric = sfric;
  m->mtype = mtype; m->mot_k = mot_k; m->mot_admit = mot_admit; m->mot_vmax = mot_vmax;
  m->mot_vmin = mot_vmin; m->mot_gear = mot_gear; m->mot_inertia = mot_inertia;
  m->brk_f = brk_f; m->brk_t = brk_t;
  m->nshape = nshape; m->shape_link = shape_link; m->shape_voff = shape_voff; m->shape_foff = shape_foff;
  m->shape_slide_mode = shape_slide_mode; m->shape_slide_vel = shape_slide_vel; m->shape_slide_axis = shape_slide_axis;
  m->verts = verts; m->planes = planes;
  m->npair = npair; m->pair_shape = pair_shape; m->pair_ci = pair_ci;
  m->nci = w->nci+1; m->ci_type = ci_type; m->ci_sf = ci_sf; m->ci_kf = ci_kf;
  m->ci_k = ci_k; m->ci_l = ci_l; m->ci_e = ci_e; m->ci_v = ci_v;
  m->ncand = ncand; m->cand_pair = cand_pair; m->cand_side = cand_side; m->cand_vert = cand_vert;
  w->built = 1;
  return 0;
}
