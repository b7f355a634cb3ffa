#include "pnr_dyn_oracle.h"
int orc_dyn_available(void) { return 0; }
