/* boxdrop.c - the scenario of the reference's example/chain/boxdrop_test.c written against include/roki_fd_amd.h:
 * boxes released one above the other over the rigid floor - they land ON EACH OTHER, because registration pairs every
 * box with every other one and rkCDPairChainUnreg (called per box, as the reference's driver does at :37) removes only a
 * chain's own pairs, of which a one-link box has none.  The reference's random start poses (zRandF, :30-35) are replaced
 * by fixed ones of the same kind (stacked heights 0.1 + 0.15 i, offsets within +-0.1, slightly tilted) so that the run is
 * reproducible.  Every rkFDUpdate runs on the GPU.
 *
 * build: gcc -O2 -Iinclude examples/boxdrop.c -Lroki-fd_amd -lrkfd_amd -Wl,-rpath,$PWD/roki-fd_amd -o boxdrop
 * usage: ./boxdrop [nbox] [steps] [model dir] [mlcp|vert|volume]      (the reference's driver selects Volume)
 */
#include <stdlib.h>
#include <string.h>
#include "roki_fd_amd.h"

#define DT    0.001
#define NMAX  9

int main(int argc, char *argv[])
{
  rkFD fd;
  int i, k, n, steps;
  rkFDCell *cell[NMAX];
  zVec dis[NMAX];
  char name[BUFSIZ];
  const char *dir = argc > 3 ? argv[3] : "models";

  rkFDCreate( &fd );
  snprintf( name, sizeof(name), "%s/contactinfo.ztk", dir );
  if( !rkFDContactInfoScanFile( &fd, name ) ) return 1;

  n = argc > 1 ? atoi( argv[1] ) : 2;
  steps = argc > 2 ? atoi( argv[2] ) : 400;
  if( n > NMAX ) n = NMAX;
  for( i=0; i<n; i++ ){
    snprintf( name, sizeof(name), "%s/box.ztk", dir );
    if( !( cell[i] = rkFDChainRegFile( &fd, name ) ) ) return 1;
    dis[i] = zVecAlloc( rkChainJointSize( rkFDCellChain(cell[i]) ) );
    zVecElemNC(dis[i],0) = 0.012*i;
    zVecElemNC(dis[i],1) = -0.009*i;
    zVecElemNC(dis[i],2) = 0.1 + i*0.15;
    zVecElemNC(dis[i],3) = zDeg2Rad( 0.5*(i+1) );
    zVecElemNC(dis[i],4) = zDeg2Rad( -0.375*(i+1) );
    zVecElemNC(dis[i],5) = 0.0;      /* (no yaw: vertex collision sees no edge-edge contact, a yawed box falls through an equal one) */
    rkFDChainSetDis( cell[i], dis[i] );
    rkCDPairChainUnreg( rkFDCDBase(&fd.cd), rkFDCellChain(cell[i]) );
  }
  snprintf( name, sizeof(name), "%s/floor.ztk", dir );
  if( !rkFDChainRegFile( &fd, name ) ) return 1;

  rkFDODE2Assign( &fd, Regular );
  rkFDODE2AssignRegular( &fd, RKG );
  rkFDPrpSetDT( &fd, DT );
  if( argc > 4 && strcmp( argv[4], "vert" ) == 0 ) rkFDSetSolver( &fd, Vert );
  else if( argc > 4 && strcmp( argv[4], "volume" ) == 0 ) rkFDSetSolver( &fd, Volume );
  else rkFDSetSolver( &fd, MLCP );

  rkFDUpdateInit( &fd );
  if( rkFDStatus( &fd ) != 0 ) return 2;
  for( k=0; k<steps; k++ ){
    rkFDUpdate( &fd );
    if( rkFDStatus( &fd ) != 0 ) return 2;
  }
  rkFDUpdateDestroy( &fd );
  printf( "t %.6f\n", rkFDTime(&fd) );
  for( i=0; i<n; i++ ){
    rkChainGetJointDisAll( rkFDCellChain(cell[i]), dis[i] );
    printf( "box %d", i );
    for( k=0; k<6; k++ ) printf( " %.12e", zVecElemNC(dis[i],k) );
    printf( "\n" );
    zVecFree( dis[i] );
  }
  rkFDDestroy( &fd );
  return 0;
}
