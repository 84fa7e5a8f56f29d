/* parse_csv.h -- forwarding header: the reference's include name, our consolidated ABI. */
#include "grtcode_hip_api.h"
