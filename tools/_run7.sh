for d in 0 1 2 4 6 7 3; do echo "DBG=$d"; DLCO_SYM_DBG=$d timeout -k 10 100 ./tools/kern_time prod 2>&1 | grep SYMMETRIC; done
