// Counts the floating-point operations of the oracle: its C source is compiled here with `double`
// replaced by a number type whose arithmetic operators bump a counter.  Test infrastructure only.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static unsigned long long g_flops = 0;
struct cd {
  double v;
  cd() = default;
  cd(double x) : v(x) {}
  cd(int x) : v(x) {}
  explicit operator double() const { return v; }
  explicit operator int() const { return (int)v; }
  explicit operator bool() const { return v != 0; }
  cd &operator+=(cd o){ g_flops++; v += o.v; return *this; }
  cd &operator-=(cd o){ g_flops++; v -= o.v; return *this; }
  cd &operator*=(cd o){ g_flops++; v *= o.v; return *this; }
  cd &operator/=(cd o){ g_flops++; v /= o.v; return *this; }
  cd operator-() const { return cd(-v); }
};
#define BIN(op) \
  static inline cd operator op(cd a, cd b){ g_flops++; return cd(a.v op b.v); } \
  static inline cd operator op(cd a, double b){ g_flops++; return cd(a.v op b); } \
  static inline cd operator op(double a, cd b){ g_flops++; return cd(a op b.v); } \
  static inline cd operator op(cd a, int b){ g_flops++; return cd(a.v op b); } \
  static inline cd operator op(int a, cd b){ g_flops++; return cd(a op b.v); }
BIN(+) BIN(-) BIN(*) BIN(/)
#define CMP(op) \
  static inline bool operator op(cd a, cd b){ return a.v op b.v; } \
  static inline bool operator op(cd a, double b){ return a.v op b; } \
  static inline bool operator op(double a, cd b){ return a op b.v; } \
  static inline bool operator op(cd a, int b){ return a.v op b; } \
  static inline bool operator op(int a, cd b){ return a op b.v; }
CMP(<) CMP(>) CMP(<=) CMP(>=) CMP(==) CMP(!=)
#define FN1(f) static inline cd f(cd a){ g_flops++; return cd(std::f(a.v)); }
FN1(sqrt) FN1(sin) FN1(cos) FN1(exp) FN1(acos) FN1(asin) FN1(tan) FN1(atan)
static inline cd fabs(cd a){ return cd(std::fabs(a.v)); }
static inline cd atan2(cd a, cd b){ g_flops++; return cd(std::atan2(a.v, b.v)); }
static inline cd fmax(cd a, cd b){ return a.v > b.v ? a : b; }
static inline cd fmin(cd a, cd b){ return a.v < b.v ? a : b; }
#define double cd
extern "C" {
#include "rkfd_oracle.c"
}
#undef double
extern "C" unsigned long long rkfdOracleFlops(void){ return g_flops; }
extern "C" void rkfdOracleFlopsReset(void){ g_flops = 0; }
