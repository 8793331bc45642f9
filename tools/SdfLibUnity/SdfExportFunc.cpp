// libSdfLibUnity.so: the reference's Unity plugin interface (src/tools/SdfLibUnity/SdfExportFunc.h:16-58) on Linux, on top of
// libsdfhip.so.  The whole implementation is the header; this is the one translation unit that instantiates it.
#include "SdfLib/SdfExportFunc.h"
