/* rkfd_sim.c - host C API (include/roki_fd_amd.h): the rkFD object of the reference
 * (reference src/rkfd_sim.c) re-built as a thin shim over the flattened world + the GPU batch.
 * No dynamics is computed here; rkFDUpdate launches the device step with a batch of one.
 */
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <time.h>
#include "roki_fd_amd.h"
#include "rkfd_world.h"

typedef struct {
  rkfdWorld world;
  rkfdBatch *batch;
  double *motor_in;   /* [nlink] of the world, host copy */
  int dirty;          /* host state newer than device state */
  int status;
  int ncell;
  int bad_ode;        /* an integrator without a device path was asked for (rkFDODE2Assign*) */
} rkFDImpl;

typedef struct { int kind; int max_rigid; } rkFDSolverPrpAMD;

#define IMPL(fd) ( (rkFDImpl *)(fd)->impl )

/* ---- zVec --------------------------------------------------------------------------- */
zVec zVecAlloc(int size)
{
  zVec v = (zVec)malloc( sizeof(zVecStruct) );
  if( !v ) return NULL;
  v->size = size;
  v->buf = (double *)calloc( size > 0 ? size : 1, sizeof(double) );
  if( !v->buf ){ free( v ); return NULL; }
  return v;
}
void zVecFree(zVec v){ if( v ){ free( v->buf ); free( v ); } }
void zVecFreeAtOnce(int n, ...)
{
  va_list ap;
  va_start( ap, n );
  while( n-- > 0 ) zVecFree( va_arg( ap, zVec ) );
  va_end( ap );
}
void zRandInit(void){ srand( (unsigned)time( NULL ) ); }
/* uniform random number in [min,max] (ZM's zRandF, used by reference example/chain/boxdrop_test.c:30-32) */
double zRandF(double min, double max){ return min + ( max - min )*( (double)rand()/(double)RAND_MAX ); }
void zVecFPrint(FILE *fp, zVec v)
{
  int i;
  if( !v ){ fprintf( fp, "(null vector)\n" ); return; }
  fprintf( fp, "%d (", v->size );
  for( i=0; i<v->size; i++ ) fprintf( fp, " %.10g", v->buf[i] );
  fprintf( fp, " )\n" );
}

/* ---- chain / joint views ------------------------------------------------------------ */
int rkChainJointSize(rkChain *c){ return c->ndof; }
int rkChainLinkNum(rkChain *c){ return c->nlink; }
void rkChainGetJointDisAll(rkChain *c, zVec dis)
{
  memcpy( dis->buf, c->fd->dis->buf + c->dof_off, sizeof(double)*c->ndof );
}
void rkChainGetJointVelAll(rkChain *c, zVec vel)
{
  memcpy( vel->buf, c->fd->vel->buf + c->dof_off, sizeof(double)*c->ndof );
}
static int joint_dof_off(rkJoint *j, int *dof)
{
  rkFDImpl *im = IMPL( j->chain->fd );
  rkfdChainDesc *cd = im->world.chain[j->chain->id];
  int i, off = j->chain->dof_off;
  for( i=0; i<j->link; i++ ) off += rkfd_joint_dof( cd->link[i].jtype );
  *dof = rkfd_joint_dof( cd->link[j->link].jtype );
  return off;
}
void rkJointGetDis(rkJoint *j, double *dis)
{
  int dof, off = joint_dof_off( j, &dof );
  memcpy( dis, j->chain->fd->dis->buf + off, sizeof(double)*dof );
}
void rkJointGetVel(rkJoint *j, double *vel)
{
  int dof, off = joint_dof_off( j, &dof );
  memcpy( vel, j->chain->fd->vel->buf + off, sizeof(double)*dof );
}
void rkJointMotorSetInput(rkJoint *j, double *input)
{
  rkFDImpl *im = IMPL( j->chain->fd );
  if( im->motor_in ) im->motor_in[j->chain->link_off + j->link] = *input;
}

/* ---- solver plugin table -------------------------------------------------------------- */
void rkFDSolverInit(rkFDSolver *solver)
{
  solver->prp = NULL; solver->com = NULL;
  solver->t = 0; solver->fdprp = NULL; solver->cd = NULL; solver->fd = NULL;
}
void rkFDSolverReset(rkFDSolver *solver)
{
  if( solver->prp ) free( solver->prp );
  solver->prp = NULL; solver->com = NULL;
}
void rkFDSolverDestroy(rkFDSolver *solver)
{
  rkFDSolverReset( solver );
  rkFDSolverInit( solver );
}

static void defci_vert(rkFDSolver *s, rkContactInfo *ci)
{ /* reference src/rkfd_vert.c:340-348 */
  (void)s; memset( ci, 0, sizeof(*ci) );
  ci->type = RKFD_CONTACT_RIGID; ci->k = 1000.0; ci->l = 1.0; ci->sf = 0.5; ci->kf = 0.3;
}
static void defci_mlcp(rkFDSolver *s, rkContactInfo *ci)
{ /* reference src/rkfd_mlcp.c:301-310 */
  (void)s; memset( ci, 0, sizeof(*ci) );
  ci->type = RKFD_CONTACT_RIGID; ci->sf = 0.5; ci->kf = 0.3; ci->k = 1000.0; ci->l = 1.0;
}
static bool solver_init(rkFDSolver *s)
{
  rkFD *fd = s->fd;
  rkFDImpl *im = IMPL( fd );
  rkFDSolverPrpAMD *p = (rkFDSolverPrpAMD *)s->prp;
  if( im->batch ){ rkfdBatchDestroy( im->batch ); im->batch = NULL; }
  if( !rkFDBuildModel( fd ) ) return false;
  {
    /* rigid contact capacity of the single-instance path: 16 vertices; under the Vert plugin one
     * pyramid face per lane bounds it to 64 / pyramid (8 for the default 8-face pyramid) */
    int mr = p->max_rigid;
    if( p->kind == RKFD_SOLVER_VERT && fd->prp.pyramid > 0 && mr*fd->prp.pyramid > 64 ) mr = 64/fd->prp.pyramid;
    im->batch = rkfdBatchCreate( &im->world.model, 1, 0, mr );
  }
  if( !im->batch ){
    fprintf( stderr, "rkfd: %s\n", rkfdHipLastError() );
    return false;
  }
  return true;
}
static void solver_colchk(rkFDSolver *s, bool doUpRef){ (void)s; (void)doUpRef; /* fused into _update */ }
static bool solver_update(rkFDSolver *s, bool doUpRef)
{
  rkFDImpl *im = IMPL( s->fd );
  if( !im->batch ) return false;
  return rkfdBatchEval( im->batch, doUpRef ? 1 : 0, NULL ) == 0;
}
static void solver_update_ref(rkFDSolver *s){ (void)s; /* prev driving torque is committed on the device */ }
static void solver_destroy(rkFDSolver *s)
{
  rkFDImpl *im = IMPL( s->fd );
  if( im->batch ){ rkfdBatchDestroy( im->batch ); im->batch = NULL; }
}
static void defci_volume(rkFDSolver *s, rkContactInfo *ci)
{ /* reference src/rkfd_volume.c:942-950 */
  (void)s; memset( ci, 0, sizeof(*ci) );
  ci->type = RKFD_CONTACT_RIGID; ci->k = 1000.0; ci->l = 0.001; ci->sf = 0.5; ci->kf = 0.3;
}
static rkFDSolverCom rkfd_solver_Volume = { defci_volume, solver_init, solver_colchk, solver_update, solver_update_ref, solver_destroy };
static rkFDSolverCom rkfd_solver_Vert = { defci_vert, solver_init, solver_colchk, solver_update, solver_update_ref, solver_destroy };
static rkFDSolverCom rkfd_solver_MLCP = { defci_mlcp, solver_init, solver_colchk, solver_update, solver_update_ref, solver_destroy };

static rkFDSolver *solver_create(rkFDSolver *s, int kind, rkFDSolverCom *com)
{
  rkFDSolverPrpAMD *p = (rkFDSolverPrpAMD *)malloc( sizeof(rkFDSolverPrpAMD) );
  if( !p ) return NULL;
  p->kind = kind; p->max_rigid = 16;
  s->prp = p; s->com = com;
  return s;
}
rkFDSolver *rkFDSolverCreate_Vert(rkFDSolver *s){ return solver_create( s, RKFD_SOLVER_VERT, &rkfd_solver_Vert ); }
rkFDSolver *rkFDSolverCreate_MLCP(rkFDSolver *s){ return solver_create( s, RKFD_SOLVER_MLCP, &rkfd_solver_MLCP ); }
/* Volume (reference src/rkfd_volume.c): rigid pairs of convex shapes go by their intersection volumes on the device
 * (csrc/device/rkfd_dev_volume.h); max_rigid counts PAIRS in collision at once here (at most 10).  Seven by default: pairs x
 * ( 1 + contact-plane conditions per pair ) is bounded by 64, and with seven pairs every pair keeps the eight conditions a pair
 * of boxes can produce (ten pairs would leave five, which the hand of an arm pressed flat against a brick exceeds) */
rkFDSolver *rkFDSolverCreate_Volume(rkFDSolver *s)
{
  rkFDSolver *r = solver_create( s, RKFD_SOLVER_VOLUME, &rkfd_solver_Volume );
  if( r ) ( (rkFDSolverPrpAMD *)r->prp )->max_rigid = 7;
  return r;
}

/* ---- rkFD --------------------------------------------------------------------------- */
rkFD *rkFDCreate(rkFD *fd)
{
  rkFDImpl *im = (rkFDImpl *)calloc( 1, sizeof(rkFDImpl) );
  if( !im ) return NULL;
  memset( fd, 0, sizeof(rkFD) );
  fd->impl = im;
  rkfdWorldInit( &im->world );
  fd->t = 0.0;
  /* rkFDPrpInit (reference src/rkfd_property.c:10-18) */
  fd->prp.dt = 0.001; fd->prp.pyramid = 8; fd->prp.friction_weight = 100; fd->prp.max_iter = 10; fd->prp.vel_eps = 1.0e-8;
  fd->cd.world = &im->world;
  rkFDSolverInit( &fd->solver );
  fd->solver.fdprp = &fd->prp; fd->solver.cd = &fd->cd; fd->solver.fd = fd;
  rkFDSetSolver( fd, Vert );   /* default collision solver (reference src/rkfd_sim.c:52) */
  return fd;
}

void rkFDDestroy(rkFD *fd)
{
  rkFDImpl *im = IMPL( fd );
  rkFDCell *c, *n;
  if( !im ) return;
  if( im->batch ) rkfdBatchDestroy( im->batch );
  rkFDSolverDestroy( &fd->solver );
  for( c=fd->list; c; c=n ){ n = c->next; free( c->chain.joint ); free( c->shape ); free( c ); }
  zVecFree( fd->dis ); zVecFree( fd->vel ); zVecFree( fd->acc );
  rkfdWorldDestroy( &im->world );
  free( im->motor_in );
  free( im );
  memset( fd, 0, sizeof(rkFD) );
}

/* _rkFDCellPush (reference src/rkfd_sim.c:188-209): takes ownership of cd */
static rkFDCell *cell_push(rkFD *fd, rkfdChainDesc *cd)
{
  rkFDImpl *im = IMPL( fd );
  rkFDCell *lc, **tail;
  zVec nd, nv, na;
  int id, i;

  if( !( lc = (rkFDCell *)calloc( 1, sizeof(rkFDCell) ) ) ){ rkfdChainDescFree( cd ); return NULL; }
  lc->chain.link_off = rkfdWorldChainLinkOffset( &im->world, im->world.nchain );
  lc->chain.dof_off  = rkfdWorldChainDofOffset( &im->world, im->world.nchain );
  if( ( id = rkfdWorldAddChain( &im->world, cd ) ) < 0 ){ rkfdChainDescFree( cd ); free( lc ); return NULL; }
  lc->chain.fd = fd; lc->chain.id = id; lc->chain.nlink = cd->nlink; lc->chain.ndof = cd->ndof;
  lc->chain.joint = (rkJoint *)calloc( cd->nlink, sizeof(rkJoint) );
  for( i=0; i<cd->nlink; i++ ){ lc->chain.joint[i].chain = &lc->chain; lc->chain.joint[i].link = i; }
  lc->shape = (zShape3D *)calloc( cd->nshape > 0 ? cd->nshape : 1, sizeof(zShape3D) );
  for( i=0; i<cd->nshape; i++ ){ lc->shape[i].cell = lc; lc->shape[i].index = i; }
  for( tail=&fd->list; *tail; tail=&(*tail)->next );
  *tail = lc;
  /* grow the packed joint state (reference src/rkfd_sim.c:79-110) */
  nd = zVecAlloc( fd->size + cd->ndof ); nv = zVecAlloc( fd->size + cd->ndof ); na = zVecAlloc( fd->size + cd->ndof );
  if( !nd || !nv || !na ){ zVecFree( nd ); zVecFree( nv ); zVecFree( na ); rkFDDestroy( fd ); return NULL; }
  if( fd->size ){
    memcpy( nd->buf, fd->dis->buf, sizeof(double)*fd->size );
    memcpy( nv->buf, fd->vel->buf, sizeof(double)*fd->size );
  }
  zVecFree( fd->dis ); zVecFree( fd->vel ); zVecFree( fd->acc );
  fd->dis = nd; fd->vel = nv; fd->acc = na;
  fd->size += cd->ndof;
  im->motor_in = (double *)realloc( im->motor_in, sizeof(double)*( lc->chain.link_off + cd->nlink ) );
  for( i=0; i<cd->nlink; i++ ) im->motor_in[lc->chain.link_off+i] = 0.0;
  im->dirty = 1; im->ncell++;
  return lc;
}

rkFDCell *rkFDChainRegFile(rkFD *fd, char filename[])
{
  rkfdChainDesc *cd;
  if( !( cd = rkfdChainReadZTK( filename ) ) ) return NULL;
  return cell_push( fd, cd );
}

/* registers a CLONE of a chain (reference src/rkfd_sim.c:211-222).  Chains exist here only as the
 * views of registered cells, so the argument is the chain of a cell of this or another rkFD. */
rkFDCell *rkFDChainReg(rkFD *fd, rkChain *chain)
{
  rkfdChainDesc *cd;
  if( !chain || !chain->fd ) return NULL;
  if( !( cd = rkfdChainDescClone( IMPL( chain->fd )->world.chain[chain->id] ) ) ) return NULL;
  return cell_push( fd, cd );
}

/* reference src/rkfd_sim.c:237-255: the cell leaves the list, the packed state shrinks
 * (_rkFDAllocJointStatePop :112-140) and its shapes leave the collision registry.  As in the
 * reference, not between rkFDUpdateInit and rkFDUpdateDestroy. */
bool rkFDChainUnreg(rkFD *fd, rkFDCell *cell)
{
  rkFDImpl *im = IMPL( fd );
  rkFDCell **pp, *lc, *c;
  int nd, nl, k;

  for( pp=&fd->list; *pp && *pp != cell; pp=&(*pp)->next );
  if( !( lc = *pp ) ) return false;
  nd = lc->chain.ndof; nl = lc->chain.nlink;
  if( fd->size - nd > 0 ){
    zVec d2 = zVecAlloc( fd->size - nd ), v2 = zVecAlloc( fd->size - nd ), a2 = zVecAlloc( fd->size - nd );
    if( !d2 || !v2 || !a2 ){ zVecFree( d2 ); zVecFree( v2 ); zVecFree( a2 ); rkFDDestroy( fd ); return false; }
    memcpy( d2->buf, fd->dis->buf, sizeof(double)*lc->chain.dof_off );
    memcpy( v2->buf, fd->vel->buf, sizeof(double)*lc->chain.dof_off );
    k = fd->size - lc->chain.dof_off - nd;
    memcpy( d2->buf + lc->chain.dof_off, fd->dis->buf + lc->chain.dof_off + nd, sizeof(double)*k );
    memcpy( v2->buf + lc->chain.dof_off, fd->vel->buf + lc->chain.dof_off + nd, sizeof(double)*k );
    zVecFree( fd->dis ); zVecFree( fd->vel ); zVecFree( fd->acc );
    fd->dis = d2; fd->vel = v2; fd->acc = a2;
  } else {
    zVecFree( fd->dis ); zVecFree( fd->vel ); zVecFree( fd->acc );
    fd->dis = fd->vel = fd->acc = NULL;
  }
  fd->size -= nd;
  k = rkfdWorldChainLinkOffset( &im->world, im->world.nchain ) - lc->chain.link_off - nl;
  memmove( im->motor_in + lc->chain.link_off, im->motor_in + lc->chain.link_off + nl, sizeof(double)*k );
  for( c=lc->next; c; c=c->next ){ c->chain.id--; c->chain.link_off -= nl; c->chain.dof_off -= nd; }
  *pp = lc->next;
  rkfdWorldRemoveChain( &im->world, lc->chain.id );
  free( lc->chain.joint ); free( lc->shape ); free( lc );
  if( im->batch ){ rkfdBatchDestroy( im->batch ); im->batch = NULL; }
  im->dirty = 1; im->ncell--;
  return true;
}

void rkFDChainSetDis(rkFDCell *lc, zVec dis)
{
  rkFD *fd = lc->chain.fd;
  memcpy( fd->dis->buf + lc->chain.dof_off, dis->buf, sizeof(double)*lc->chain.ndof );
  IMPL( fd )->dirty = 1;
}
void rkFDChainSetVel(rkFDCell *lc, zVec vel)
{
  rkFD *fd = lc->chain.fd;
  memcpy( fd->vel->buf + lc->chain.dof_off, vel->buf, sizeof(double)*lc->chain.ndof );
  IMPL( fd )->dirty = 1;
}

bool rkFDContactInfoScanFile(rkFD *fd, char filename[])
{
  return rkfdWorldSetContactInfo( &IMPL( fd )->world, filename ) == 0;
}

void rkCDPairChainUnreg(rkFDCD *cd, rkChain *chain)
{
  rkfdWorldPairChainUnreg( (rkfdWorld *)cd->world, chain->id );
}

const rkfdModel *rkFDBuildModel(rkFD *fd)
{
  rkFDImpl *im = IMPL( fd );
  rkFDSolverPrpAMD *p = (rkFDSolverPrpAMD *)fd->solver.prp;
  rkfdWorld *w = &im->world;
  w->cidef.type = fd->cidef.type; w->cidef.sf = fd->cidef.sf; w->cidef.kf = fd->cidef.kf;
  w->cidef.k = fd->cidef.k; w->cidef.l = fd->cidef.l; w->cidef.e = fd->cidef.e; w->cidef.v = fd->cidef.v;
  w->model.dt = fd->prp.dt;
  w->model.friction_weight = fd->prp.friction_weight;
  w->model.max_iter = fd->prp.max_iter;
  w->model.pyramid = fd->prp.pyramid;
  w->model.solver = p ? p->kind : RKFD_SOLVER_VERT;
  if( rkfdWorldBuild( w ) < 0 ) return NULL;
  return &w->model;
}

rkfdBatch *rkFDBatchCreate(rkFD *fd, int batch, int device, int max_rigid)
{
  const rkfdModel *m = rkFDBuildModel( fd );
  if( !m ) return NULL;
  return rkfdBatchCreate( m, batch, device, max_rigid );
}

static int sync_to_device(rkFD *fd)
{
  rkFDImpl *im = IMPL( fd );
  if( !im->batch ) return -1;
  if( im->dirty ){
    if( rkfdBatchSetState( im->batch, fd->dis->buf, fd->vel->buf ) < 0 ) return -1;
    im->dirty = 0;
  }
  return rkfdBatchSetMotorInput( im->batch, im->motor_in );
}

static void report(rkFD *fd)
{
  rkFDImpl *im = IMPL( fd );
  im->status = im->batch ? rkfdBatchStatus( im->batch, NULL ) : -1;
  if( im->status != 0 ) fprintf( stderr, "rkfd: %s\n", rkfdHipLastError() );
}

/* rkFDODE2Assign / rkFDODE2AssignRegular: only "Regular" + "RKG" have a device path (see roki_fd_amd.h) */
void rkfd_ode2_assign(rkFD *fd, const char *what, const char *type)
{
  rkFDImpl *im = IMPL( fd );
  const int ok = strcmp( what, "rkFDODE2Assign" ) == 0 ? strcmp( type, "Regular" ) == 0 : strcmp( type, "RKG" ) == 0;
  if( !ok ){
    fprintf( stderr, "rkfd: %s( fd, %s ): this integrator has no device path (only Regular + RKG, the reference's default); updates are refused until a supported one is assigned\n", what, type );
    im->bad_ode |= strcmp( what, "rkFDODE2Assign" ) == 0 ? 1 : 2;
    im->status = -2;
  } else {
    im->bad_ode &= strcmp( what, "rkFDODE2Assign" ) == 0 ? ~1 : ~2;
    if( !im->bad_ode && im->status == -2 ) im->status = 0;
  }
}

void rkFDFK(rkFD *fd, zVec dis)
{
  if( !dis || dis->size < fd->size ) return;
  memcpy( fd->dis->buf, dis->buf, sizeof(double)*fd->size );
  IMPL( fd )->dirty = 1;
}
void rkFDUpdateRate(rkFD *fd, zVec vel, zVec acc)
{
  if( vel && vel->size >= fd->size ) memcpy( fd->vel->buf, vel->buf, sizeof(double)*fd->size );
  if( acc && acc->size >= fd->size ) memcpy( fd->acc->buf, acc->buf, sizeof(double)*fd->size );
  IMPL( fd )->dirty = 1;
}
void rkFDUpdateFKRate(rkFD *fd){ IMPL( fd )->dirty = 1; }

void rkFDFPrintZTK(FILE *fp, rkFD *fd)
{
  rkFDImpl *im = IMPL( fd );
  rkFDCell *lc;
  for( lc=fd->list; lc; lc=lc->next )
    rkfdChainWriteZTK( fp, im->world.chain[lc->chain.id], fd->dis ? fd->dis->buf + lc->chain.dof_off : NULL );
}
void rkFDPrint(rkFD *fd){ rkFDFPrintZTK( stdout, fd ); }

/* slide mode through the reference's names (handles: see roki_fd_amd.h) */
zShape3D *rkFDCellShape(rkFDCell *cell, int i)
{
  rkFDImpl *im = IMPL( cell->chain.fd );
  return ( i >= 0 && i < im->world.chain[cell->chain.id]->nshape ) ? &cell->shape[i] : NULL;
}
int rkFDCellShapeNum(rkFDCell *cell){ return IMPL( cell->chain.fd )->world.chain[cell->chain.id]->nshape; }
static rkfdShape *cd_shape(rkCDCell *c)
{
  rkFDImpl *im = IMPL( c->cell->chain.fd );
  im->world.built = 0; im->dirty = 1;
  return &im->world.chain[c->cell->chain.id]->shape[c->index];
}
void rkFDCDCellSetSlideMode(rkCDCell *cell, bool mode){ if( cell ) cd_shape( cell )->slide_mode = mode ? 1 : 0; }
void rkFDCDCellSetSlideVel(rkCDCell *cell, double vel){ if( cell ) cd_shape( cell )->slide_vel = vel; }
void rkFDCDCellSetSlideAxis(rkCDCell *cell, zVec3D *axis){ if( cell && axis ) memcpy( cd_shape( cell )->slide_axis, axis->e, sizeof(double)*3 ); }
rkCDCell *rkFDShape3DGetCDCell(rkFD *fd, zShape3D *shape){ return ( shape && shape->cell && shape->cell->chain.fd == fd ) ? shape : NULL; }
rkCDCell *rkFDShape3DSetSlideMode(rkFD *fd, zShape3D *shape, bool mode)
{ rkCDCell *c = rkFDShape3DGetCDCell( fd, shape ); if( c ) rkFDCDCellSetSlideMode( c, mode ); return c; }
rkCDCell *rkFDShape3DSetSlideVel(rkFD *fd, zShape3D *shape, double vel)
{ rkCDCell *c = rkFDShape3DGetCDCell( fd, shape ); if( c ) rkFDCDCellSetSlideVel( c, vel ); return c; }
rkCDCell *rkFDShape3DSetSlideAxis(rkFD *fd, zShape3D *shape, zVec3D *axis)
{ rkCDCell *c = rkFDShape3DGetCDCell( fd, shape ); if( c ) rkFDCDCellSetSlideAxis( c, axis ); return c; }

void rkFDUpdateInit(rkFD *fd)
{
  rkFDImpl *im = IMPL( fd );
  if( im->bad_ode ){ fprintf( stderr, "rkfd: rkFDUpdateInit refused: an integrator without a device path is assigned\n" ); im->status = -2; return; }
  im->dirty = 1;
  if( !rkFDSolverUpdateInit( &fd->solver ) ){ im->status = -1; return; }
  if( sync_to_device( fd ) < 0 || rkfdBatchUpdateInit( im->batch, NULL ) < 0 ){
    fprintf( stderr, "rkfd: %s\n", rkfdHipLastError() );
    im->status = -1;
    return;
  }
  report( fd );
  rkfdBatchGetState( im->batch, NULL, NULL, fd->acc->buf );
}

rkFD *rkFDUpdate(rkFD *fd)
{
  rkFDImpl *im = IMPL( fd );
  if( im->bad_ode ){ fprintf( stderr, "rkfd: rkFDUpdate refused: an integrator without a device path is assigned\n" ); im->status = -2; return fd; }
  if( !im->batch ){
    fprintf( stderr, "rkfd: rkFDUpdate called without a device batch (rkFDUpdateInit failed or missing)\n" );
    im->status = -1;
    return fd;
  }
  if( sync_to_device( fd ) < 0 || rkfdBatchUpdate( im->batch, 1, NULL ) < 0 ){
    fprintf( stderr, "rkfd: %s\n", rkfdHipLastError() );
    im->status = -1;
    return fd;
  }
  report( fd );
  rkfdBatchGetState( im->batch, fd->dis->buf, fd->vel->buf, fd->acc->buf );
  fd->t += rkFDDT( fd );
  return fd;
}

void rkFDUpdateDestroy(rkFD *fd)
{
  if( fd->solver.com ) rkFDSolverUpdateDestroy( &fd->solver );
}

rkFD *rkFDSolve(rkFD *fd)
{
  rkFDUpdateInit( fd );
  rkFDUpdate( fd );
  rkFDUpdateDestroy( fd );
  return fd;
}

int rkFDStatus(rkFD *fd){ return IMPL( fd )->status; }

/* ---- flat entry points for FFI users that drive the world builder directly ----------- */
rkfdWorld *rkfdWorldCreate(void)
{
  rkfdWorld *w = (rkfdWorld *)malloc( sizeof(rkfdWorld) );
  if( w ) rkfdWorldInit( w );
  return w;
}
void rkfdWorldFree(rkfdWorld *w){ if( w ){ rkfdWorldDestroy( w ); free( w ); } }
int rkfdWorldRegFile(rkfdWorld *w, const char *filename)
{
  rkfdChainDesc *c = rkfdChainReadZTK( filename );
  int id;
  if( !c ) return -1;
  if( ( id = rkfdWorldAddChain( w, c ) ) < 0 ) rkfdChainDescFree( c );
  return id;
}
void rkfdWorldSetPrp(rkfdWorld *w, double dt, double friction_weight, int max_iter, int solver)
{
  w->model.dt = dt; w->model.friction_weight = friction_weight; w->model.max_iter = max_iter; w->model.solver = solver;
  /* default contact info follows the solver, as rkFDSetSolver does (reference src/rkfd_vert.c:340-348, src/rkfd_mlcp.c:301-310:
   * relaxation 1.0; src/rkfd_volume.c:961-969: 0.001) */
  w->cidef.type = RKFD_CONTACT_RIGID; w->cidef.k = 1000.0; w->cidef.l = solver == RKFD_SOLVER_VOLUME ? 0.001 : 1.0; w->cidef.sf = 0.5; w->cidef.kf = 0.3;
  w->built = 0;
}
/* rkFDCDCellSetSlideMode / Vel / Axis (reference src/rkfd_sim.c:384-401) on shape number `shape` of a chain
 * (its order in the ZTK file); axis in the link frame.  Returns 0, -1 for an unknown chain / shape. */
int rkfdWorldSetSlide(rkfdWorld *w, int chain, int shape, int mode, double vel, const double axis[3])
{
  rkfdShape *sh;
  if( chain < 0 || chain >= w->nchain || shape < 0 || shape >= w->chain[chain]->nshape ) return -1;
  sh = &w->chain[chain]->shape[shape];
  sh->slide_mode = mode ? 1 : 0; sh->slide_vel = vel;
  if( axis ){ sh->slide_axis[0] = axis[0]; sh->slide_axis[1] = axis[1]; sh->slide_axis[2] = axis[2]; }
  w->built = 0;
  return 0;
}
/* the same through the rkFD mirror: the role of rkFDShape3DSetSlideMode / Vel / Axis (reference src/rkfd_sim.c:412-440),
 * the shape named by its number in the cell's chain */
bool rkFDCellSetSlide(rkFDCell *cell, int shape, bool mode, double vel, const double axis[3])
{
  rkFDImpl *im = IMPL( cell->chain.fd );
  if( rkfdWorldSetSlide( &im->world, cell->chain.id, shape, mode, vel, axis ) != 0 ) return false;
  im->dirty = 1;
  return true;
}
/* rkFDPrpSetPyramid for the flat loader: faces of the Vert plugin's friction pyramid */
void rkfdWorldSetPyramid(rkfdWorld *w, int pyramid){ w->model.pyramid = pyramid; w->built = 0; }
const rkfdModel *rkfdWorldModel(rkfdWorld *w)
{
  if( !w->built && rkfdWorldBuild( w ) < 0 ) return NULL;
  return &w->model;
}
int rkfdWorldWriteZTK(const rkfdWorld *w, int chain, const char *filename, const double *dis)
{
  FILE *fp;
  if( chain < 0 || chain >= w->nchain || !( fp = fopen( filename, "w" ) ) ) return -1;
  rkfdChainWriteZTK( fp, w->chain[chain], dis );
  fclose( fp );
  return 0;
}
int rkfdWorldChainInitDis(const rkfdWorld *w, int chain, double *dis)
{
  if( chain < 0 || chain >= w->nchain ) return -1;
  memcpy( dis, w->chain[chain]->init_dis, sizeof(double)*w->chain[chain]->ndof );
  return w->chain[chain]->ndof;
}
