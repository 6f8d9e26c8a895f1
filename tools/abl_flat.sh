for args in "--n 160 --cin 512 --cout 1536 --hw 16 --ks 1 --plain" "--n 160 --cin 512 --cout 512 --hw 16 --ks 1 --plain" "--n 160 --cin 1024 --cout 512 --hw 16 --ks 1 --plain" "--n 160 --cin 256 --cout 128 --hw 64 --ks 1 --plain" "--n 160 --cin 512 --cout 512 --hw 16 --ks 1"; do
  echo "== $args"; bash tools/ablate_conv.sh "$args" "8 16 32 64"
done
