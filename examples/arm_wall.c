/* arm_wall.c - the scenario of the reference's example/chain/arm_wall_test.c written against include/roki_fd_amd.h with the models
 * of this repository: an arm under a joint-level PD controller (every other step, as in the reference's control()) swings its hand
 * into a column of bricks that hang on each other by BREAKABLE FLOAT joints (models/wall.ztk: the structure and thresholds of the
 * reference's wall.ztk); the joints give way, the bricks come loose.  As in the reference's driver, rkCDPairChainUnreg is NOT
 * called for the wall (its bricks are cells of one chain and do collide once they are loose); it is called for the arm here, whose
 * links carry one shape only.  Every rkFDUpdate runs on the GPU.
 *
 * build: gcc -O2 -Iinclude examples/arm_wall.c -Lroki-fd_amd -lrkfd_amd -Wl,-rpath,$PWD/roki-fd_amd -o arm_wall
 * usage: ./arm_wall [steps] [model dir] [mlcp|volume]      (the reference's driver selects Volume)
 */
#include <stdlib.h>
#include <string.h>
#include "roki_fd_amd.h"

#define DT   0.001
#define DTC  0.002
#define KP   60.0
#define KD   3.0

static const double target[2] = { 1.9, 0.12 };     /* yaw (sweeping the hand through the column), shoulder pitch */

static void control(rkFDCell *cell)
{
  int i;
  double dis, vel, e;
  rkJoint *joint;

  for( i=0; i<2; i++ ){
    joint = rkChainLinkJoint( rkFDCellChain(cell), i );
    rkJointGetDis( joint, &dis );
    rkJointGetVel( joint, &vel );
    e = -KP*( dis - target[i] ) - KD*vel;
    rkJointMotorSetInput( joint, &e );
  }
}

int main(int argc, char *argv[])
{
  rkFD fd;
  rkFDCell *cell[2];
  zVec dis[2];
  char name[BUFSIZ];
  const int steps = argc > 1 ? atoi( argv[1] ) : 300;
  const char *dir = argc > 2 ? argv[2] : "models";
  double t_cnt;
  int k, i;

  rkFDCreate( &fd );
  snprintf( name, sizeof(name), "%s/contactinfo.ztk", dir );
  if( !rkFDContactInfoScanFile( &fd, name ) ) return 1;
  snprintf( name, sizeof(name), "%s/arm_revroot.ztk", dir );
  if( !( cell[0] = rkFDChainRegFile( &fd, name ) ) ) return 1;
  snprintf( name, sizeof(name), "%s/wall.ztk", dir );
  if( !( cell[1] = rkFDChainRegFile( &fd, name ) ) ) return 1;
  snprintf( name, sizeof(name), "%s/floor.ztk", dir );
  if( !rkFDChainRegFile( &fd, name ) ) return 1;

  dis[0] = zVecAlloc( rkChainJointSize( rkFDCellChain(cell[0]) ) );
  zVecElemNC(dis[0],0) = 1.40;      /* yaw: the hand just short of the column */
  zVecElemNC(dis[0],1) = 0.12;
  rkFDChainSetDis( cell[0], dis[0] );
  rkCDPairChainUnreg( rkFDCDBase(&fd.cd), rkFDCellChain(cell[0]) );
  dis[1] = zVecAlloc( rkChainJointSize( rkFDCellChain(cell[1]) ) );

  rkFDODE2Assign( &fd, Regular );
  rkFDODE2AssignRegular( &fd, RKG );
  rkFDPrpSetDT( &fd, DT );
  if( argc > 3 && strcmp( argv[3], "volume" ) == 0 ) rkFDSetSolver( &fd, Volume );
  else rkFDSetSolver( &fd, MLCP );

  rkFDUpdateInit( &fd );
  if( rkFDStatus( &fd ) != 0 ) return 2;
  t_cnt = rkFDTime( &fd );
  for( k=0; k<steps; k++ ){
    if( t_cnt <= rkFDTime( &fd ) + 1.0e-9 ){
      control( cell[0] );
      t_cnt += DTC;
    }
    rkFDUpdate( &fd );
    if( rkFDStatus( &fd ) != 0 ) return 2;
  }
  rkFDUpdateDestroy( &fd );
  printf( "t %.6f\n", rkFDTime( &fd ) );
  for( i=0; i<2; i++ ){
    rkChainGetJointDisAll( rkFDCellChain(cell[i]), dis[i] );
    printf( "%s", i == 0 ? "arm" : "wall" );
    for( k=0; k<rkChainJointSize( rkFDCellChain(cell[i]) ); k++ ) printf( " %.12e", zVecElemNC(dis[i],k) );
    printf( "\n" );
    zVecFree( dis[i] );
  }
  rkFDDestroy( &fd );
  return 0;
}
