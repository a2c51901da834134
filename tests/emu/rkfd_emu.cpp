/* rkfd_emu.cpp - DEVELOPMENT / TEST HARNESS ONLY.
 *
 * Runs the device code of roki-fd_amd/csrc/rkfd_device.h on the CPU by emulating one
 * 64-lane wavefront with 64 host threads and barriers, so that the kernel LOGIC can be
 * checked against the oracle in the no-GPU test tier.  It is not a fallback: nothing in the
 * product library links or loads it, and the C-ABI fails loudly without a GPU.
 */
#include <barrier>
#include <thread>
#include <vector>
#include <cstring>
#include <cstdlib>

#define RKFD_EMU
/* RKFD_W (instances per wavefront, rkfd_devmodel.h) comes from the build: librkfd_emu.so has 1, librkfd_emu_w2.so has 2 - the
 * 64 threads are then two instances of 32 lanes, and the cross-lane helpers act within the caller's half, as on the device */
#ifndef RKFD_W
#define RKFD_W 1
#endif
#define EMU_WL ( 64/RKFD_W )
static thread_local int t_tid;                /* thread = lane of the wavefront, 0 .. 63 */
#define t_lane ( t_tid & ( EMU_WL-1 ) )       /* lane within the instance */
#define t_half0 ( t_tid & ~( EMU_WL-1 ) )     /* first thread of the caller's instance */
/* one barrier per INSTANCE: the two instances of a wavefront exchange nothing, and what one of them branches on (its contact
 * count ...) the other does not share - on the device the execution mask serialises such branches, here the halves simply run
 * on their own */
static std::barrier<> *g_bars[RKFD_W];
#define g_bar ( g_bars[t_tid/EMU_WL] )
static double g_xd[64];
static unsigned long long g_xb[64];

int rkfd_emu_lane(void){ return t_lane; }
int rkfd_emu_half(void){ return t_tid/EMU_WL; }
void rkfd_emu_sync(void){ g_bar->arrive_and_wait(); }
static std::barrier<> *g_wavebar;            /* every thread that runs in this wavefront (both instances, or the only live one) */
void rkfd_emu_sync_wave(void){ g_wavebar->arrive_and_wait(); }
double rkfd_emu_g8sum(double x)
{
  g_xd[t_tid] = x;
  g_bar->arrive_and_wait();
  /* same association as the DPP butterfly: (l^1), then (l^2), then mirrored half */
  const int b = t_tid & ~7, l = t_tid & 7;
  auto q = [&](int k){ int base = k & ~3; double s[4]; for( int i=0; i<4; i++ ) s[i] = g_xd[b+base+i];
                       int j = k & 3; double p1 = s[j] + s[j^1]; double p2 = s[j^2] + s[(j^2)^1]; return p1 + p2; };
  double r = q( l ) + q( 7-l );
  g_bar->arrive_and_wait();
  return r;
}
double rkfd_emu_g8bcast(double x, int k)
{
  g_xd[t_tid] = x;
  g_bar->arrive_and_wait();
  double r = g_xd[( t_tid & ~7 ) + k];
  g_bar->arrive_and_wait();
  return r;
}
double rkfd_emu_bcast(double x, int src)
{
  g_xd[t_tid] = x;
  g_bar->arrive_and_wait();
  double r = g_xd[t_half0 + src];
  g_bar->arrive_and_wait();
  return r;
}
/* sum / minimum over the wave, the same in every lane; the sum in the association of the device code: the 8-lane butterfly
 * of rkfd_emu_g8sum as lane 0 of each group sees it, then the eight groups in order */
double rkfd_emu_wsum(double x)
{
  g_xd[t_tid] = x;
  g_bar->arrive_and_wait();
  double r = 0;
  for( int b=t_half0; b<t_half0+EMU_WL; b+=8 ){
    auto q = [&](int k){ int base = k & ~3; double s[4]; for( int i=0; i<4; i++ ) s[i] = g_xd[b+base+i];
                         int j = k & 3; double p1 = s[j] + s[j^1]; double p2 = s[j^2] + s[(j^2)^1]; return p1 + p2; };
    r += q( 0 ) + q( 7 );
  }
  g_bar->arrive_and_wait();
  return r;
}
double rkfd_emu_wmin(double x)
{
  g_xd[t_tid] = x;
  g_bar->arrive_and_wait();
  double r = g_xd[t_half0];
  for( int i=1; i<EMU_WL; i++ ) if( g_xd[t_half0+i] < r ) r = g_xd[t_half0+i];
  g_bar->arrive_and_wait();
  return r;
}
unsigned long long rkfd_emu_ballot(int pred)
{
  g_xb[t_tid] = pred ? 1ull : 0ull;
  g_bar->arrive_and_wait();
  unsigned long long m = 0;
  for( int i=0; i<EMU_WL; i++ ) m |= g_xb[t_half0 + i] << i;
  g_bar->arrive_and_wait();
  return m;
}

#include "rkfd_device.h"
#include "rkfd_devmodel_host.h"

extern "C" int rkfd_emu_run(const rkfdModel *m, int max_rigid, rkfdDevState *st, int mode, int nsteps)
{
  rkfdDevModelHost h;
  char err[256];
  if( rkfd_devmodel_build_w( m, max_rigid, 8/RKFD_W, &h, err, sizeof(err) ) < 0 ) return -1;
  /* the harness keeps the state arrays at the boundary convention (anchors in model link frames) */
  if( h.ncand > 0 ) rkfd_ref_to_device( &h, st->cv_ref, (size_t)st->batch*h.ncand );
  std::vector<char> lds( RKFD_W*h.lds_bytes + h.dm.lds_shared + 64 );
  int errflag = 0;
  for( int b=0; b<st->batch; b+=RKFD_W ){
    std::barrier<> bar0( EMU_WL ), bar1( EMU_WL );
    g_bars[0] = &bar0; if( RKFD_W > 1 ) g_bars[RKFD_W-1] = &bar1;
    int nrun = 0;
    for( int l=0; l<64; l++ ) nrun += b + l/EMU_WL < st->batch;
    std::barrier<> wbar( nrun );
    g_wavebar = &wbar;
    /* LDS is NOT cleared on the GPU: poison it (all ones: NaN as a double, -1 as an int), so that a read of storage nobody wrote
     * shows here instead of depending on what the previous kernel on the box left behind */
    std::memset( lds.data(), 0xFF, lds.size() );
    std::vector<std::thread> th;
    for( int l=0; l<64; l++ ){
      /* the half of an odd batch's last wavefront that has no instance of its own: on the device its lanes run in lockstep with
       * the other half on a copy of that half's instance and store nothing; here the halves are independent threads (per-instance
       * barriers), so the stand-in is simply not run - it would read the state the live half is writing */
      if( b + l/EMU_WL >= st->batch ) continue;
      th.emplace_back( [&, l](){ t_tid = l;
        /* the instance of this thread's half; a half beyond the batch stands in with the instance before it and stores nothing */
        int bi = b + l/EMU_WL;
        const bool live = bi < st->batch;
        if( !live ) bi -= 1;
        char *base = lds.data() + ( l/EMU_WL )*h.lds_bytes;
        char *shared = lds.data() + RKFD_W*h.lds_bytes;      /* (the world's static tables, once per wavefront: rkfdDevModel.lds_shared) */
        /* the variant the C-ABI would launch: the one carrying the Vert QP only for worlds that need it */
        if( h.dm.vol_np > 0 ) rkfd_instance<false, 2, false>( h.dm, *st, bi, base, mode, nsteps, &errflag, live, shared );
        else if( h.dm.vert_rigid ) rkfd_instance<false, 1, false>( h.dm, *st, bi, base, mode, nsteps, &errflag, live, shared );
        else if( h.dm.ma_packed ) rkfd_instance<false, 0, true>( h.dm, *st, bi, base, mode, nsteps, &errflag, live, shared );
        else rkfd_instance<false, 0, false>( h.dm, *st, bi, base, mode, nsteps, &errflag, live, shared ); } );
    }
    for( auto &t : th ) t.join();
  }
  if( h.ncand > 0 ) rkfd_ref_to_model( &h, st->cv_ref, (size_t)st->batch*h.ncand );
  rkfd_devmodel_free( &h );
  return errflag;
}
