/* shim: see oracle/ref_shim/cudaemu.h */
#pragma once
#include "../cudaemu.h"
