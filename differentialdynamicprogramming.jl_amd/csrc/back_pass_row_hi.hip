// second translation unit of back_pass_row.hip: the padded sizes NP = 10, 12, 14 (compiled beside the first)
#define DDP_ROW_PART 1
#include "back_pass_row.hip"
