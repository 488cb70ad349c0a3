// pipeline.h — ring of N buffers handed from player to player in a fixed order: same contract as the
// reference's util Pipeline (include/UtilLib/pipeline.h:31-131): player p may take the next slot only after
// player p-1 released it; RequestBuffer blocks, RequestBufferTimeoutable gives up after timeout_secs
// (0 = a single try) and returns nullptr.  Waiting uses a condition variable instead of the reference's
// 100 us sleep loop.
#ifndef REBVO_AMD_HOST_PIPELINE_H
#define REBVO_AMD_HOST_PIPELINE_H

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <stdexcept>
#include <vector>

namespace rebvo {

template <class OPipe>
class Pipeline {
    const unsigned CircSize, PlayerNum;
    std::vector<OPipe> CircBuff;
    std::vector<unsigned> CircPlayer;   // last player that released each slot
    std::vector<unsigned> PlayerPos;    // slot each player currently holds
    std::mutex mut;
    std::condition_variable cv;

    bool ready(int PlayerId, unsigned next) const { return CircPlayer[next] == (PlayerId + PlayerNum - 1) % PlayerNum; }

public:
    Pipeline(unsigned CircularSize, unsigned PlayerNumber)
        : CircSize(CircularSize), PlayerNum(PlayerNumber), CircBuff(CircularSize),
          CircPlayer(CircularSize, PlayerNumber - 1), PlayerPos(PlayerNumber, 0) {}

    void ReleaseBuffer(unsigned PlayerId) {
        {
            std::lock_guard<std::mutex> locker(mut);
            CircPlayer[PlayerPos[PlayerId]] = PlayerId;
        }
        cv.notify_all();
    }

    // A player that holds more than one buffer (it requested the next before it was done with the one before) releases the older
    // ones by address, oldest first.  Not in the reference: its players hold one buffer at a time.
    void ReleaseBufferAt(unsigned PlayerId, const OPipe *buf) {
        {
            std::lock_guard<std::mutex> locker(mut);
            CircPlayer[(size_t)(buf - CircBuff.data())] = PlayerId;
        }
        cv.notify_all();
    }

    OPipe &RequestBuffer(int PlayerId) {
        std::unique_lock<std::mutex> lk(mut);
        const unsigned next = (PlayerPos[PlayerId] + 1) % CircSize;
        cv.wait(lk, [&] { return ready(PlayerId, next); });
        PlayerPos[PlayerId] = next;
        return CircBuff[next];
    }

    OPipe *RequestBufferTimeoutable(int PlayerId, double timeout_secs = 0) {
        std::unique_lock<std::mutex> lk(mut);
        const unsigned next = (PlayerPos[PlayerId] + 1) % CircSize;
        if (timeout_secs > 0) {
            if (!cv.wait_for(lk, std::chrono::duration<double>(timeout_secs), [&] { return ready(PlayerId, next); })) return nullptr;
        } else if (!ready(PlayerId, next)) {   // a single try is a look at the ring, not a timed wait that has already expired (tens of
            return nullptr;                    // microseconds in the C library: a batch group polls a thousand rings per pass)
        }
        PlayerPos[PlayerId] = next;
        return &CircBuff[next];
    }

    // construction-time access only (not safe once the players run), as in the reference
    OPipe &operator[](unsigned inx) {
        if (inx >= CircSize) throw std::out_of_range("Circular size out of range");
        return CircBuff[inx];
    }
    unsigned Size() const { return CircSize; }
    typename std::vector<OPipe>::iterator begin() { return CircBuff.begin(); }
    typename std::vector<OPipe>::iterator end() { return CircBuff.end(); }
};

}  // namespace rebvo
#endif
