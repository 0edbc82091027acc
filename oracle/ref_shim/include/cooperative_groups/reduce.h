/* shim: cooperative_groups/reduce.h is included by the reference but nothing from it is used */
#pragma once
