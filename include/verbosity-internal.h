/* verbosity-internal.h -- forwarding header: the reference's include name, our consolidated ABI. */
#include "debug.h"
