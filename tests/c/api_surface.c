/* the reference's public names that need no GPU (include/roki_fd_amd.h): integrator assignment, the Volume plugin table,
 * slide mode through rkFDShape3DSetSlide*, rkFDFK / rkFDUpdateRate, rkFDFPrintZTK.  Prints facts, the Python test asserts. */
#include <stdio.h>
#include <string.h>
#include "roki_fd_amd.h"

int main(int argc, char *argv[])
{
  rkFD fd, fd2;
  rkFDCell *box, *chain, *back;
  char path[1024];
  zVec dis, vel, acc;
  zVec3D axis = { { 0.0, 0.0, 1.0 } };
  const rkfdModel *m;
  FILE *fp;
  int i;

  if( argc < 3 ) return 1;
  rkFDCreate( &fd );
  snprintf( path, sizeof(path), "%s/contactinfo.ztk", argv[1] ); rkFDContactInfoScanFile( &fd, path );
  snprintf( path, sizeof(path), "%s/box.ztk", argv[1] ); box = rkFDChainRegFile( &fd, path );
  snprintf( path, sizeof(path), "%s/arm_revroot.ztk", argv[1] ); chain = rkFDChainRegFile( &fd, path );
  /* integrators: the reference's drivers assign Regular + RKG */
  rkFDODE2Assign( &fd, Regular );
  rkFDODE2AssignRegular( &fd, RKG );
  printf( "status after Regular + RKG: %d\n", rkFDStatus( &fd ) );
  rkFDODE2AssignRegular( &fd, RK4 );
  printf( "status after RK4: %d\n", rkFDStatus( &fd ) );
  rkFDUpdate( &fd );                                   /* refused, no device touched */
  printf( "time after the refused update: %g\n", rkFDTime( &fd ) );
  rkFDODE2AssignRegular( &fd, RKG );
  printf( "status after RKG again: %d\n", rkFDStatus( &fd ) );
  /* plugin tables */
  rkFDSetSolver( &fd, Volume );
  printf( "Volume default relaxation %g\n", fd.cidef.l );
  rkFDSetSolver( &fd, MLCP );
  /* slide mode through the reference's names */
  printf( "box shapes %d\n", rkFDCellShapeNum( box ) );
  printf( "slide cell %s\n", rkFDShape3DSetSlideMode( &fd, rkFDCellShape( box, 0 ), true ) ? "found" : "missing" );
  rkFDShape3DSetSlideVel( &fd, rkFDCellShape( box, 0 ), 0.25 );
  rkFDShape3DSetSlideAxis( &fd, rkFDCellShape( box, 0 ), &axis );
  m = rkFDBuildModel( &fd );
  printf( "slide mode %d vel %g axis %g %g %g\n", m->shape_slide_mode[0], m->shape_slide_vel[0], m->shape_slide_axis[0], m->shape_slide_axis[1], m->shape_slide_axis[2] );
  /* packed state through rkFDFK / rkFDUpdateRate */
  dis = zVecAlloc( fd.size ); vel = zVecAlloc( fd.size ); acc = zVecAlloc( fd.size );
  for( i=0; i<fd.size; i++ ){ zVecElemNC( dis, i ) = 0.01*( i+1 ); zVecElemNC( vel, i ) = -0.02*( i+1 ); }
  rkFDFK( &fd, dis ); rkFDUpdateRate( &fd, vel, acc ); rkFDUpdateFKRate( &fd );
  printf( "dis[7] %g vel[7] %g\n", zVecElemNC( fd.dis, 7 ), zVecElemNC( fd.vel, 7 ) );
  /* rkFDPrint: an rkFD holding the arm alone is written and read back */
  if( !rkFDChainUnreg( &fd, box ) ) return 2;
  zVecElemNC( fd.dis, 2 ) = 0.125;
  fp = fopen( argv[2], "w" ); rkFDFPrintZTK( fp, &fd ); fclose( fp );
  rkFDCreate( &fd2 );
  back = rkFDChainRegFile( &fd2, argv[2] );
  printf( "read back: %s, size %d\n", back ? "ok" : "failed", back ? rkChainJointSize( rkFDCellChain( back ) ) : -1 );
  (void)chain;
  zVecFree( dis ); zVecFree( vel ); zVecFree( acc );
  rkFDDestroy( &fd2 ); rkFDDestroy( &fd );
  return 0;
}
