export DIR_HEAD=bc7363a
bash tools/profile_round.sh r06_a_prof
bash tools/profile_four_in_flight.sh
bash tools/pmc_fwd_sq.sh r06_a_sq conv_as_kernel conv_pipe_kernel stream1x1_kernel
