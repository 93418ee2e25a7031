// Host-side probe of conv_common.h: choose_tile_order (tests/test_tile_order.py): reads "B H W Cin Cout K tiles_m tiles_n es" lines from stdin,
// prints 1 (column-major) or 0 (row-major) per line.  No device code runs.
#include "conv_common.h"
#include <cstdio>
int main() {
    long long B, H, W, Cin, Cout, K, tm, tn, es;
    while (scanf("%lld %lld %lld %lld %lld %lld %lld %lld %lld", &B, &H, &W, &Cin, &Cout, &K, &tm, &tn, &es) == 9) {
        dir::convk::ConvArgs a{};
        a.B = (int)B; a.H = (int)H; a.W = (int)W; a.Cin = (int)Cin; a.Cout = (int)Cout; a.K = (int)K; a.tiles_m = (int)tm; a.tiles_n = (int)tn;
        a.flags = 4;
        dir::convk::choose_tile_order(a, (int)es);
        printf("%d\n", (a.flags & dir::convk::CONV_COL_MAJOR) ? 1 : 0);
    }
    return 0;
}
