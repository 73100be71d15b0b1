for v in 21 11 12; do
  echo "variant $v"; WGS_EM_CODED_VARIANT=$v timeout -k 10 200 python tools/check_codes.py 10000000 1000 10 | python -c "import sys,json; d=json.load(sys.stdin); print(d['fit_coded'], d['fit_identical'])"
done
