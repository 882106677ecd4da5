for c in 1 4; do export KIDMP_CPW=$c; echo "CPW=$c"; bash tools/pass_times.sh config3 config2; done
