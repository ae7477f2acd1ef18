set -e
cd $GRAFT_REPO_ROOT
RTC_CXXFLAGS="-DRTC_STAMPS" python raytracer-challenge_amd/build.py --force > /dev/null 2>&1
python tools/phase_shares.py ${1:-100} ${2:-flat} $3 $4
