// dw_main.cc -- `dw`: drop-in for the reference's command line (src/main.cc:1-4).
#include "dw_cli.h"

int main(int argc, const char *const argv[]) { return dw::dw_main(argc, argv); }
