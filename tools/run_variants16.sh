for v in "$@"; do YART_LIB=$v python tools/trace_ab.py 0 2>&1 | grep -v "rep=0"; done
