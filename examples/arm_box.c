/* arm_box.c - the scenario of the reference's example/chain/arm_box_test.c written against include/roki_fd_amd.h with the
 * models of this repository: an arm under a joint-level PD controller that runs every other step (rkJointGetDis / GetVel /
 * MotorSetInput, like the reference's control()), a free box and the rigid floor, MLCP plugin.  Every rkFDUpdate runs on
 * the GPU; the controller runs on the host between steps, as in the reference.
 *
 * build: gcc -O2 -Iinclude examples/arm_box.c -Lroki-fd_amd -lrkfd_amd -Wl,-rpath,$PWD/roki-fd_amd -o arm_box
 * usage: ./arm_box [steps] [model dir] [mlcp|volume]      (the reference's driver selects Volume)
 */
#include <stdlib.h>
#include <string.h>
#include "roki_fd_amd.h"

#define DT   0.001
#define DTC  0.002
#define KP   40.0
#define KD   2.0

static const double target[4] = { 0.3, 0.9, 0.02, -0.2 };     /* yaw, shoulder, forearm extension, wrist */

/* the PD law of the reference's control(), per driven joint (the arm's yaw has a DC motor: the input is a voltage,
 * its shoulder a torque motor) */
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
  snprintf( name, sizeof(name), "%s/box.ztk", dir );
  if( !( cell[1] = rkFDChainRegFile( &fd, name ) ) ) return 1;
  snprintf( name, sizeof(name), "%s/floor.ztk", dir );
  if( !rkFDChainRegFile( &fd, name ) ) return 1;

  dis[0] = zVecAlloc( rkChainJointSize( rkFDCellChain(cell[0]) ) );
  zVecElemNC(dis[0],1) = 0.6;
  rkFDChainSetDis( cell[0], dis[0] );
  /* the arm's OWN pairs go (self-collision off); it goes on touching the box and the floor - exactly where the reference's
   * driver has the call (example/chain/arm_box_test.c:49, after arm, box and floor are registered) */
  rkCDPairChainUnreg( rkFDCDBase(&fd.cd), rkFDCellChain(cell[0]) );
  dis[1] = zVecAlloc( rkChainJointSize( rkFDCellChain(cell[1]) ) );
  zVecElemNC(dis[1],0) = 0.25; zVecElemNC(dis[1],1) = 0.05; zVecElemNC(dis[1],2) = 0.05 - 1.0e-5;
  rkFDChainSetDis( cell[1], dis[1] );

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
    printf( "%s", i == 0 ? "arm" : "box" );
    for( k=0; k<rkChainJointSize( rkFDCellChain(cell[i]) ); k++ ) printf( " %.12e", zVecElemNC(dis[i],k) );
    printf( "\n" );
    zVecFree( dis[i] );
  }
  rkFDDestroy( &fd );
  return 0;
}
