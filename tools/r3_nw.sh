cd /root/repo 2>/dev/null || cd $GRAFT_REPO_ROOT
for nw in 8 4; do
  touch gpismap_amd/csrc/ongpis_train.hip
  make -C gpismap_amd/csrc EXTRA="-DK3_T0_NW=$nw" > /tmp/mk.log 2>&1 || { tail -20 /tmp/mk.log; exit 1; }
  echo "== K3_T0_NW=$nw"
  python -m pytest tests/test_gpu_ongpis.py -q -x 2>&1 | tail -1
  python3 tools/update_profile.py 8 2>/dev/null | grep "^frame [234567]" | sed 's/| pts.*//; s/.*K3 device/K3/; s/)//' | tr '\n' ' '; echo
  python3 tools/update_profile.py 8 2>/dev/null | grep "^frame [234567]" | sed 's/| pts.*//; s/.*K3 device/K3/; s/)//' | tr '\n' ' '; echo
  for cfg in "350 256" "350 512" "200 512"; do python3 tools/k3_bench.py $cfg 2>&1 | grep "^N="; done
done
