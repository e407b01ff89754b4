// include/ltxhip.h of this repository (pass -Xcc -I<repo>/include)
#include <ltxhip.h>
