#include "../../include/asr_hip.h"
extern "C" int asr_version(void) { return 1; }
