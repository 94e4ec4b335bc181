# Entry points under the names the reference repository uses for them (its Makefile: stark_parser, generate_proof, unit_test,
# integration_test), mapped onto this backend. Everything else lives in aero_amd/csrc/Makefile, oracle/Makefile.
PYTHON ?= python3
PROOF ?= proofs/fibair_2p10.bin

.PHONY: all build bin/stark_parser generate_proof verify_proof unit_test integration_test bench clean

all: build

# libaero_stark.so for gfx950 (hipcc cross-compiles without a GPU), bin/stark_parser, the test oracle
build:
	$(PYTHON) -c "import __graft_entry__ as g; g.build()"

# the reference builds its Rust parser into this path; here it is the same command line over the library
bin/stark_parser:
	$(MAKE) -C aero_amd/csrc ../../bin/stark_parser

# reference: `cargo run -p miden_proof_generator` -> proofs/fib.bin. Here: the 2^10-row Fibonacci trace of the built-in AIR, proven
# on the GPU, written as the same bincode ProofData container (needs an MI355X)
generate_proof: build
	mkdir -p proofs
	$(PYTHON) -m aero_amd prove --width 2 --log-n 10 --out $(PROOF)

verify_proof:
	$(PYTHON) -m aero_amd verify $(PROOF)

# reference: protostar unit / integration tests of the Cairo verifier. Here: the CPU suite (oracle vs the reference's golden
# vectors, host logic, ABI) and the GPU parity suite
unit_test:
	$(PYTHON) -m pytest tests -q -m "not gpu"

integration_test:
	$(PYTHON) -m pytest tests -q -m gpu

bench:
	$(PYTHON) bench.py

clean:
	$(MAKE) -C aero_amd/csrc clean
	$(MAKE) -C oracle clean
	rm -rf bin
